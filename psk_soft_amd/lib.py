"""ctypes binding of libpsk_soft_hip.so (include/psk_soft_hip.h).

The shared library is the product: HIP kernels for gfx950 behind a C ABI.  This module
only marshals arguments.  It never computes anything itself and has no CPU fallback: if
the library is missing, or no MI355X is visible, creation raises.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpsk_soft_hip.so")

DEVICE_NONE = -1
OK = 0
NOOP, NORMAL = 0, 1

STATUS_NAMES = {
    0: "PSK_SOFT_OK",
    1: "PSK_SOFT_ERR_INVALID_ARG",
    2: "PSK_SOFT_ERR_NO_DEVICE",
    3: "PSK_SOFT_ERR_HIP",
    4: "PSK_SOFT_ERR_LIMIT",
    5: "PSK_SOFT_ERR_UNSUPPORTED",
    6: "PSK_SOFT_ERR_CAPACITY",
}

PROP_NAMES = ("samplesPerBaud", "constelationSize", "numAvg", "phaseAvg", "differentialDecoding", "resetState")


class Props(ctypes.Structure):
    _fields_ = [
        ("samplesPerBaud", ctypes.c_uint16),
        ("constelationSize", ctypes.c_uint16),
        ("numAvg", ctypes.c_uint32),
        ("phaseAvg", ctypes.c_uint16),
        ("differentialDecoding", ctypes.c_uint8),
        ("resetState", ctypes.c_uint8),
    ]


class Limits(ctypes.Structure):
    _fields_ = [
        ("max_window_samples", ctypes.c_uint32),
        ("max_phase_avg", ctypes.c_uint32),
        ("max_packet_complex", ctypes.c_uint32),
    ]


class Packet(ctypes.Structure):
    _fields_ = [
        ("data", ctypes.c_void_p),
        ("n_floats", ctypes.c_uint64),
        ("sri_xdelta", ctypes.c_double),
        ("sri_mode", ctypes.c_int32),
        ("sriChanged", ctypes.c_uint8),
        ("inputQueueFlushed", ctypes.c_uint8),
        ("present", ctypes.c_uint8),
        ("reserved", ctypes.c_uint8),
    ]


class Output(ctypes.Structure):
    _fields_ = [
        ("soft", ctypes.c_void_p),
        ("bits", ctypes.c_void_p),
        ("phase", ctypes.c_void_p),
        ("sampleIndex", ctypes.c_void_p),
        ("cap_symbols", ctypes.c_uint64),
        ("ret", ctypes.c_int32),
        ("n_symbols", ctypes.c_uint64),
        ("n_bits", ctypes.c_uint64),
        ("n_sampleIndex", ctypes.c_uint64),
        ("sri_pushed", ctypes.c_int32),
        ("sri_soft_xdelta", ctypes.c_double),
        ("sri_bits_xdelta", ctypes.c_double),
        ("n_warn", ctypes.c_int32),
    ]


class Stats(ctypes.Structure):
    _fields_ = [
        ("channels_fast", ctypes.c_uint64),
        ("channels_exact_timing", ctypes.c_uint64),
        ("channels_sequential", ctypes.c_uint64),
        ("channels_guard", ctypes.c_uint64),
        ("unwrap_extra_passes", ctypes.c_uint64),
        ("unwrap_blocks", ctypes.c_uint64),
        ("timing_exact_blocks", ctypes.c_uint64),
        ("fit_chain_blocks", ctypes.c_uint64),
        ("channels_tiled", ctypes.c_uint64),
        ("channels_parallel_fit", ctypes.c_uint64),
        ("channels_parallel_fit_second_round", ctypes.c_uint64),
        ("parallel_fit_refusals", ctypes.c_uint64),
    ]


# every symbol include/psk_soft_hip.h declares
EXPORTS = (
    "psk_soft_device_alloc",
    "psk_soft_device_free",
    "psk_soft_device_upload",
    "psk_soft_device_download",
    "psk_soft_get_channel_stats",
    "psk_soft_probe_read_ms",
    "psk_soft_set_option",
    "psk_soft_host_alloc",
    "psk_soft_host_free",
    "psk_soft_abi_version",
    "psk_soft_last_error",
    "psk_soft_create",
    "psk_soft_destroy",
    "psk_soft_configure",
    "psk_soft_query",
    "psk_soft_fire_listener",
    "psk_soft_output_capacity",
    "psk_soft_process_device",
    "psk_soft_process_host",
    "psk_soft_synchronize",
    "psk_soft_join",
    "psk_soft_get_stats",
    "psk_soft_set_force_sequential",
    "psk_soft_state_bytes",
    "psk_soft_export_state",
    "psk_soft_import_state",
    "psk_soft_peek",
)


class PskSoftError(RuntimeError):
    def __init__(self, status, text):
        super().__init__("%s: %s" % (STATUS_NAMES.get(status, status), text))
        self.status = status


_lib = None


def load():
    """Load the shared library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libpsk_soft_hip.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C psk_soft_amd/csrc`); there is no CPU fallback"
        )
    L = ctypes.CDLL(LIB_PATH)
    vp, u32, u64, i32 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int
    L.psk_soft_abi_version.restype = u32
    L.psk_soft_last_error.restype = ctypes.c_char_p
    L.psk_soft_create.argtypes = [i32, u32, ctypes.POINTER(Limits), ctypes.POINTER(vp)]
    L.psk_soft_destroy.argtypes = [vp]
    L.psk_soft_configure.argtypes = [vp, u32, u32, ctypes.POINTER(Props)]
    L.psk_soft_query.argtypes = [vp, u32, ctypes.POINTER(Props)]
    L.psk_soft_fire_listener.argtypes = [vp, u32, i32]
    L.psk_soft_output_capacity.argtypes = [vp, u32, u64]
    L.psk_soft_output_capacity.restype = u64
    L.psk_soft_process_device.argtypes = [vp, u32, u32, ctypes.POINTER(Packet), ctypes.POINTER(Output), vp]
    L.psk_soft_process_host.argtypes = [vp, u32, u32, ctypes.POINTER(Packet), ctypes.POINTER(Output)]
    L.psk_soft_synchronize.argtypes = [vp]
    L.psk_soft_join.argtypes = [vp, vp]
    L.psk_soft_get_stats.argtypes = [vp, ctypes.POINTER(Stats)]
    L.psk_soft_get_channel_stats.argtypes = [vp, u32, u32, ctypes.POINTER(Stats)]
    L.psk_soft_set_force_sequential.argtypes = [vp, i32]
    L.psk_soft_set_option.argtypes = [vp, i32, i32]
    L.psk_soft_state_bytes.argtypes = [vp]
    L.psk_soft_state_bytes.restype = u64
    L.psk_soft_export_state.argtypes = [vp, u32, vp, u64]
    L.psk_soft_import_state.argtypes = [vp, u32, vp, u64]
    L.psk_soft_peek.argtypes = [vp, u32, ctypes.POINTER(u64), ctypes.POINTER(u64), ctypes.POINTER(u64)]
    L.psk_soft_host_alloc.argtypes = [ctypes.c_size_t]
    L.psk_soft_host_alloc.restype = vp
    L.psk_soft_host_free.argtypes = [vp]
    L.psk_soft_device_alloc.argtypes = [vp, ctypes.c_size_t]
    L.psk_soft_device_alloc.restype = vp
    L.psk_soft_device_free.argtypes = [vp, vp]
    L.psk_soft_device_upload.argtypes = [vp, vp, vp, ctypes.c_size_t]
    L.psk_soft_device_download.argtypes = [vp, vp, vp, ctypes.c_size_t]
    L.psk_soft_probe_read_ms.argtypes = [vp, vp, u64, i32, ctypes.POINTER(ctypes.c_double)]
    _lib = L
    return L


def host_alloc(n, dtype):
    """A numpy array of `n` elements in pinned host memory (psk_soft_host_alloc): hand its
    .ctypes.data to Handle.process_device.  Keep the returned array alive; free with host_free."""
    import numpy as np

    dt = np.dtype(dtype)
    p = load().psk_soft_host_alloc(int(n) * dt.itemsize)
    if not p:
        raise MemoryError("psk_soft_host_alloc failed: " + load().psk_soft_last_error().decode())
    buf = (ctypes.c_char * (int(n) * dt.itemsize)).from_address(p)
    arr = np.frombuffer(buf, dtype=dt)
    return arr


def host_free(arr):
    load().psk_soft_host_free(ctypes.c_void_p(arr.ctypes.data))


def _check(status):
    if status != OK:
        raise PskSoftError(status, load().psk_soft_last_error().decode("utf-8", "replace"))


class Handle:
    """A batch of channels on one GPU (psk_soft_handle_t)."""

    def __init__(self, n_channels, device=0, max_window_samples=16384, max_phase_avg=512, max_packet_complex=1 << 20):
        L = load()
        self._L = L
        self.n_channels = int(n_channels)
        self.device = device
        lim = Limits(max_window_samples, max_phase_avg, max_packet_complex)
        h = ctypes.c_void_p()
        _check(L.psk_soft_create(int(device), self.n_channels, ctypes.byref(lim), ctypes.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.psk_soft_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- properties -------------------------------------------------------------------
    def query(self, ch):
        p = Props()
        _check(self._L.psk_soft_query(self._h, ch, ctypes.byref(p)))
        return p

    def configure(self, ch0, props):
        """props: one Props or dict for every channel in [ch0, ch0+len(props))."""
        arr = (Props * len(props))()
        for i, p in enumerate(props):
            if isinstance(p, dict):
                cur = self.query(ch0 + i)
                for k in PROP_NAMES:
                    setattr(arr[i], k, int(p.get(k, getattr(cur, k))))
            else:
                arr[i] = p
        _check(self._L.psk_soft_configure(self._h, ch0, len(props), arr))

    def configure_all(self, **kw):
        cur = self.query(0)
        p = Props()
        for k in PROP_NAMES:
            setattr(p, k, int(kw.get(k, getattr(cur, k))))
        arr = (Props * self.n_channels)(*([p] * self.n_channels))
        _check(self._L.psk_soft_configure(self._h, 0, self.n_channels, arr))

    def fire_listener(self, ch, which):
        _check(self._L.psk_soft_fire_listener(self._h, ch, which))

    OPT_QPSK_SIGN_BITMAP = 1
    OPT_CONCURRENT_CLASSES = 2
    OPT_TIME_TILED = 3  # 0 never, 1 where it pays (default), 2 wherever the kernels exist
    OPT_DEFERRED_JOIN = 5  # mixed window classes: the side streams are joined by join() / synchronize(), not by every call
    OPT_PARALLEL_FIT = 4  # tiled calls: 0 fit block by block, 1 parallel fit with the second round on demand (default), 2 always

    def set_option(self, option, value):
        _check(self._L.psk_soft_set_option(self._h, int(option), int(value)))

    def set_force_sequential(self, on):
        _check(self._L.psk_soft_set_force_sequential(self._h, int(bool(on))))

    def output_capacity(self, ch, n_complex):
        return int(self._L.psk_soft_output_capacity(self._h, ch, int(n_complex)))

    def peek(self, ch):
        a, b, c = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        _check(self._L.psk_soft_peek(self._h, ch, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return {"ring_len": a.value, "index": b.value, "fit_len": c.value}

    # -- processing -------------------------------------------------------------------
    def process_device(self, ch0, pkts, outs, stream=None):
        """pkts / outs: ctypes arrays of Packet / Output holding DEVICE pointers."""
        _check(self._L.psk_soft_process_device(self._h, ch0, len(pkts), pkts, outs, ctypes.c_void_p(stream or 0)))

    def process_host(self, ch0, packets):
        """packets: list (one per channel from ch0) of None (no packet) or dict with
        data (float32 interleaved I/Q), xdelta, and optional mode / sriChanged /
        inputQueueFlushed.  Returns one dict per channel with the four output streams."""
        n = len(packets)
        pk = (Packet * n)()
        out = (Output * n)()
        keep = []
        bufs = []
        for i, p in enumerate(packets):
            if p is None:
                pk[i].present = 0
                bufs.append(None)
                continue
            data = np.ascontiguousarray(p["data"], dtype=np.float32)
            keep.append(data)
            pk[i].data = data.ctypes.data
            pk[i].n_floats = data.size
            pk[i].sri_xdelta = float(p["xdelta"])
            pk[i].sri_mode = int(p.get("mode", 1))
            pk[i].sriChanged = int(bool(p.get("sriChanged", False)))
            pk[i].inputQueueFlushed = int(bool(p.get("inputQueueFlushed", False)))
            pk[i].present = 1
            cap = self.output_capacity(ch0 + i, data.size // 2)
            soft = np.empty(2 * cap, np.float32)
            bits = np.empty(3 * cap, np.int16)
            phase = np.empty(cap, np.float32)
            sidx = np.empty(cap, np.int16)
            bufs.append((soft, bits, phase, sidx))
            out[i].soft = soft.ctypes.data
            out[i].bits = bits.ctypes.data
            out[i].phase = phase.ctypes.data
            out[i].sampleIndex = sidx.ctypes.data
            out[i].cap_symbols = cap
        _check(self._L.psk_soft_process_host(self._h, ch0, n, pk, out))
        res = []
        for i in range(n):
            o = out[i]
            if bufs[i] is None:
                soft = np.zeros(0, np.float32)
                bits = np.zeros(0, np.int16)
                phase = np.zeros(0, np.float32)
                sidx = np.zeros(0, np.int16)
            else:
                soft = bufs[i][0][: 2 * o.n_symbols]
                bits = bufs[i][1][: o.n_bits]
                phase = bufs[i][2][: o.n_symbols]
                sidx = bufs[i][3][: o.n_sampleIndex]
            res.append(
                {
                    "ret": o.ret,
                    "soft": soft,
                    "bits": bits,
                    "phase": phase,
                    "index": sidx,
                    "sri_pushed": bool(o.sri_pushed),
                    "sri_soft_xdelta": o.sri_soft_xdelta,
                    "sri_bits_xdelta": o.sri_bits_xdelta,
                    "n_warn": o.n_warn,
                }
            )
        return res

    def plan_only(self, ch0, packets):
        """Control-plane results (counts, SRI) of one call without data (DEVICE_NONE handles)."""
        n = len(packets)
        pk = (Packet * n)()
        out = (Output * n)()
        for i, p in enumerate(packets):
            if p is None:
                continue
            pk[i].n_floats = int(p["n_floats"])
            pk[i].sri_xdelta = float(p["xdelta"])
            pk[i].sri_mode = int(p.get("mode", 1))
            pk[i].sriChanged = int(bool(p.get("sriChanged", False)))
            pk[i].inputQueueFlushed = int(bool(p.get("inputQueueFlushed", False)))
            pk[i].present = 1
            out[i].cap_symbols = 1 << 62
        _check(self._L.psk_soft_process_device(self._h, ch0, n, pk, out, None))
        return [
            {
                "ret": o.ret,
                "n_symbols": o.n_symbols,
                "n_bits": o.n_bits,
                "n_sampleIndex": o.n_sampleIndex,
                "sri_pushed": bool(o.sri_pushed),
                "sri_soft_xdelta": o.sri_soft_xdelta,
                "sri_bits_xdelta": o.sri_bits_xdelta,
                "n_warn": o.n_warn,
            }
            for o in out
        ]

    def synchronize(self):
        _check(self._L.psk_soft_synchronize(self._h))

    def join(self, stream=None):
        """OPT_DEFERRED_JOIN: `stream` (a raw hipStream_t; None = the handle's own) waits for the handle's side streams."""
        _check(self._L.psk_soft_join(self._h, ctypes.c_void_p(stream) if stream else None))

    def probe_read_ms(self, dev_ptr, nbytes, reps=5):
        """Mean duration (ms) of one pure 16-byte-load pass over a device buffer (empirical read ceiling)."""
        ms = ctypes.c_double()
        _check(self._L.psk_soft_probe_read_ms(self._h, ctypes.c_void_p(dev_ptr), int(nbytes), int(reps), ctypes.byref(ms)))
        return ms.value

    def stats(self):
        s = Stats()
        _check(self._L.psk_soft_get_stats(self._h, ctypes.byref(s)))
        return {k: getattr(s, k) for k, _ in Stats._fields_}

    # -- device buffers for callers without a HIP runtime of their own (tests) --------------
    def device_alloc(self, nbytes):
        p = self._L.psk_soft_device_alloc(self._h, int(nbytes))
        if not p:
            raise PskSoftError(3, self._L.psk_soft_last_error().decode("utf-8", "replace"))
        return p

    def device_free(self, p):
        self._L.psk_soft_device_free(self._h, ctypes.c_void_p(p))

    def upload(self, dev_ptr, arr):
        arr = np.ascontiguousarray(arr)
        _check(self._L.psk_soft_device_upload(self._h, ctypes.c_void_p(dev_ptr), arr.ctypes.data, arr.nbytes))

    def download(self, dev_ptr, shape, dtype):
        out = np.empty(shape, dtype)
        _check(self._L.psk_soft_device_download(self._h, out.ctypes.data, ctypes.c_void_p(dev_ptr), out.nbytes))
        return out

    def channel_stats(self, ch0=0, nch=None):
        nch = self.n_channels - ch0 if nch is None else nch
        arr = (Stats * nch)()
        _check(self._L.psk_soft_get_channel_stats(self._h, ch0, nch, arr))
        return [{k: getattr(s, k) for k, _ in Stats._fields_} for s in arr]

    def export_state(self, ch):
        n = int(self._L.psk_soft_state_bytes(self._h))
        buf = (ctypes.c_uint8 * n)()
        _check(self._L.psk_soft_export_state(self._h, ch, buf, n))
        return bytes(buf)

    def import_state(self, ch, blob):
        buf = (ctypes.c_uint8 * len(blob)).from_buffer_copy(blob)
        _check(self._L.psk_soft_import_state(self._h, ch, buf, len(blob)))
