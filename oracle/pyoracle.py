"""ctypes binding of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import
this module; nothing under psk_soft_amd/ does.  The oracle restates
psk_soft_i::serviceFunction (reference cpp/psk_soft.cpp:346-618); see
oracle/psk_soft_oracle.h for its pinning status.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpsk_soft_oracle.so")

PROP_IDS = {
    "samplesPerBaud": 0,
    "numAvg": 1,
    "constelationSize": 2,
    "phaseAvg": 3,
    "differentialDecoding": 4,
    "resetState": 5,
}


class _Packet(ctypes.Structure):
    _fields_ = [
        ("data", ctypes.POINTER(ctypes.c_float)),
        ("n_floats", ctypes.c_size_t),
        ("xdelta", ctypes.c_double),
        ("mode", ctypes.c_int),
        ("sriChanged", ctypes.c_int),
        ("inputQueueFlushed", ctypes.c_int),
    ]


class _Result(ctypes.Structure):
    _fields_ = [
        ("ret", ctypes.c_int),
        ("soft", ctypes.POINTER(ctypes.c_float)),
        ("n_soft_floats", ctypes.c_size_t),
        ("bits", ctypes.POINTER(ctypes.c_short)),
        ("n_bits", ctypes.c_size_t),
        ("phase", ctypes.POINTER(ctypes.c_float)),
        ("n_phase", ctypes.c_size_t),
        ("index", ctypes.POINTER(ctypes.c_short)),
        ("n_index", ctypes.c_size_t),
        ("sri_pushed", ctypes.c_int),
        ("sri_soft_xdelta", ctypes.c_double),
        ("sri_bits_xdelta", ctypes.c_double),
        ("n_warn", ctypes.c_int),
    ]


def build(force=False):
    """Compile the oracle's C restatement (and prim_check) with oracle/Makefile."""
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "psk_soft_oracle.c"))
    ):
        subprocess.run(["make", "-C", _HERE, "all"], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        L.psk_oracle_create.restype = ctypes.c_void_p
        L.psk_oracle_destroy.argtypes = [ctypes.c_void_p]
        L.psk_oracle_set_property.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_int]
        L.psk_oracle_get_property.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.psk_oracle_get_property.restype = ctypes.c_uint32
        L.psk_oracle_service.argtypes = [ctypes.c_void_p, ctypes.POINTER(_Packet), ctypes.POINTER(_Result)]
        L.psk_oracle_service.restype = ctypes.c_int
        L.psk_oracle_ring_size.argtypes = [ctypes.c_void_p]
        L.psk_oracle_ring_size.restype = ctypes.c_size_t
        L.psk_oracle_index.argtypes = [ctypes.c_void_p]
        L.psk_oracle_index.restype = ctypes.c_size_t
        L.psk_oracle_phase_estimate.argtypes = [ctypes.c_void_p]
        L.psk_oracle_phase_estimate.restype = ctypes.c_float
        L.psk_oracle_fit_history.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float), ctypes.c_size_t]
        L.psk_oracle_fit_history.restype = ctypes.c_size_t
        _lib = L
    return _lib


class CallResult:
    """Outputs of one serviceFunction() call, one field per output port."""

    __slots__ = ("ret", "soft", "bits", "phase", "index", "sri_pushed", "sri_soft_xdelta", "sri_bits_xdelta", "n_warn")

    def __init__(self, r):
        self.ret = r.ret
        self.soft = np.ctypeslib.as_array(r.soft, (r.n_soft_floats,)).copy() if r.n_soft_floats else np.zeros(0, np.float32)
        self.bits = np.ctypeslib.as_array(r.bits, (r.n_bits,)).copy() if r.n_bits else np.zeros(0, np.int16)
        self.phase = np.ctypeslib.as_array(r.phase, (r.n_phase,)).copy() if r.n_phase else np.zeros(0, np.float32)
        self.index = np.ctypeslib.as_array(r.index, (r.n_index,)).copy() if r.n_index else np.zeros(0, np.int16)
        self.sri_pushed = bool(r.sri_pushed)
        self.sri_soft_xdelta = r.sri_soft_xdelta
        self.sri_bits_xdelta = r.sri_bits_xdelta
        self.n_warn = r.n_warn


class OracleComponent:
    """One psk_soft_i instance (one stream).  Properties are attributes, as on the
    sandbox component of the reference's test (tests/test_psk_soft.py:191-194);
    assigning one runs configure(): the change listener fires iff the value changed."""

    def __init__(self):
        object.__setattr__(self, "_h", lib().psk_oracle_create())

    def __del__(self):
        h = self.__dict__.get("_h")
        if h:
            lib().psk_oracle_destroy(h)
            object.__setattr__(self, "_h", None)

    def __setattr__(self, name, value):
        if name in PROP_IDS:
            old = lib().psk_oracle_get_property(self._h, PROP_IDS[name])
            new = int(value)
            lib().psk_oracle_set_property(self._h, PROP_IDS[name], new, int(old != new))
        else:
            object.__setattr__(self, name, value)

    def __getattr__(self, name):
        if name in PROP_IDS:
            return lib().psk_oracle_get_property(self._h, PROP_IDS[name])
        raise AttributeError(name)

    def configure(self, name, value, fire=True):
        lib().psk_oracle_set_property(self._h, PROP_IDS[name], int(value), int(bool(fire)))

    def service(self, data, xdelta, mode=1, sriChanged=False, inputQueueFlushed=False):
        """One serviceFunction() call on one packet of interleaved float32 I/Q."""
        data = np.ascontiguousarray(data, dtype=np.float32)
        pkt = _Packet(
            data.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
            data.size,
            float(xdelta),
            int(mode),
            int(bool(sriChanged)),
            int(bool(inputQueueFlushed)),
        )
        res = _Result()
        lib().psk_oracle_service(self._h, ctypes.byref(pkt), ctypes.byref(res))
        return CallResult(res)

    # introspection
    @property
    def ring_size(self):
        return lib().psk_oracle_ring_size(self._h)

    @property
    def index(self):
        return lib().psk_oracle_index(self._h)

    @property
    def phase_estimate(self):
        return lib().psk_oracle_phase_estimate(self._h)

    def fit_history(self):
        buf = (ctypes.c_float * 65536)()
        n = lib().psk_oracle_fit_history(self._h, buf, 65536)
        return np.array(buf[:n], dtype=np.float32)


def run_stream(comp, iq, xdelta, packet_complex=None):
    """Push a whole interleaved-I/Q float32 stream through `comp` in packets of
    `packet_complex` complex samples (None = one packet) and concatenate what the
    four output ports produced.  The first packet carries sriChanged, as a BULKIO
    source does."""
    iq = np.ascontiguousarray(iq, dtype=np.float32)
    n = iq.size // 2
    step = n if not packet_complex else int(packet_complex)
    soft, bits, phase, index = [], [], [], []
    n_sri = 0
    first = True
    pos = 0
    while True:
        cnt = min(step, n - pos) if n else 0
        r = comp.service(iq[2 * pos : 2 * (pos + cnt)], xdelta, sriChanged=first)
        first = False
        soft.append(r.soft)
        bits.append(r.bits)
        phase.append(r.phase)
        index.append(r.index)
        n_sri += int(r.sri_pushed)
        pos += cnt
        if pos >= n:
            break
    return {
        "soft": np.concatenate(soft),
        "bits": np.concatenate(bits),
        "phase": np.concatenate(phase),
        "index": np.concatenate(index),
        "n_sri": n_sri,
    }
