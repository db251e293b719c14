"""Sandbox-style driver of the C++ host class (psk_soft_amd/host/psk_soft_gpu.h).

Plays the role of `ossie.utils.sb` in the reference's component test (reference
tests/test_psk_soft.py:119-269): a component whose properties are attributes, an input
you push packets into, and four sinks you read.  Backed by libpsk_soft_host_harness.so
(in-memory ports around the real host class, which calls libpsk_soft_hip.so).
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "libpsk_soft_host_harness.so")
PROP_IDS = {"samplesPerBaud": 0, "numAvg": 1, "constelationSize": 2, "phaseAvg": 3, "differentialDecoding": 4, "resetState": 5}
PORTS = {"softDecision_dataFloat_out": 0, "bits_dataShort_out": 1, "phase_dataFloat_out": 2, "sampleIndex_dataShort_out": 3}
_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise ImportError("libpsk_soft_host_harness.so is not built (make -C psk_soft_amd/csrc)")
        L = ctypes.CDLL(_PATH)
        vp = ctypes.c_void_p
        L.psk_harness_create.restype = vp
        L.psk_harness_create.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
        L.psk_harness_destroy.argtypes = [vp]
        L.psk_harness_configure.argtypes = [vp, ctypes.c_int, ctypes.c_uint]
        L.psk_harness_push.argtypes = [vp, ctypes.POINTER(ctypes.c_float), ctypes.c_size_t, ctypes.c_double, ctypes.c_int,
                                       ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_double]
        L.psk_harness_service.argtypes = [vp]
        L.psk_harness_error.argtypes = [vp]
        L.psk_harness_error.restype = ctypes.c_char_p
        L.psk_harness_warnings.argtypes = [vp]
        L.psk_harness_port_size.argtypes = [vp, ctypes.c_int]
        L.psk_harness_port_size.restype = ctypes.c_size_t
        L.psk_harness_port_packets.argtypes = [vp, ctypes.c_int]
        L.psk_harness_port_read_f32.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_float), ctypes.c_size_t]
        L.psk_harness_port_read_f32.restype = ctypes.c_size_t
        L.psk_harness_port_read_i16.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_short), ctypes.c_size_t]
        L.psk_harness_port_read_i16.restype = ctypes.c_size_t
        L.psk_harness_port_sri.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int), ctypes.c_size_t]
        L.psk_harness_port_sri.restype = ctypes.c_size_t
        L.psk_harness_last_eos.argtypes = [vp]
        L.psk_harness_last_stream.argtypes = [vp]
        L.psk_harness_last_stream.restype = ctypes.c_char_p
        _lib = L
    return _lib


class Component:
    """comp.samplesPerBaud = 8 ... ; comp.push(...); comp.service(); comp.getData(port)."""

    def __init__(self, device=0):
        L = _load()
        err = ctypes.create_string_buffer(512)
        h = L.psk_harness_create(int(device), err, 512)
        if not h:
            raise RuntimeError(err.value.decode())
        object.__setattr__(self, "_h", ctypes.c_void_p(h))
        object.__setattr__(self, "_vals", dict(samplesPerBaud=10, numAvg=100, constelationSize=4, phaseAvg=50,
                                               differentialDecoding=0, resetState=0))

    def close(self):
        if self.__dict__.get("_h"):
            _load().psk_harness_destroy(self._h)
            object.__setattr__(self, "_h", None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __setattr__(self, name, value):
        if name in PROP_IDS:
            rc = _load().psk_harness_configure(self._h, PROP_IDS[name], int(value))
            if rc:
                raise RuntimeError(_load().psk_harness_error(self._h).decode())
            self._vals[name] = int(value)
        else:
            object.__setattr__(self, name, value)

    def __getattr__(self, name):
        if name in PROP_IDS:
            return self._vals[name]
        raise AttributeError(name)

    def push(self, data, sampleRate=None, xdelta=None, complexData=True, sriChanged=False, inputQueueFlushed=False,
             EOS=False, streamID="stream", twsec=0.0):
        """Queue one packet on dataFloat_in (sb.DataSource.push: sampleRate -> SRI.xdelta = 1/sampleRate)."""
        data = np.ascontiguousarray(data, dtype=np.float32)
        xd = (1.0 / sampleRate) if xdelta is None else xdelta
        _load().psk_harness_push(self._h, data.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), data.size, float(xd),
                                 1 if complexData else 0, int(bool(sriChanged)), int(bool(inputQueueFlushed)),
                                 int(bool(EOS)), streamID.encode(), float(twsec))

    def service(self):
        """One serviceFunction() call; returns NOOP (0) / NORMAL (1)."""
        rc = _load().psk_harness_service(self._h)
        if rc < 0:
            raise RuntimeError(_load().psk_harness_error(self._h).decode())
        return rc

    def getData(self, port):
        """Drain what `port` received (sb.DataSink.getData())."""
        L = _load()
        pid = PORTS[port]
        n = L.psk_harness_port_size(self._h, pid)
        if pid in (0, 2):
            buf = np.empty(n, np.float32)
            L.psk_harness_port_read_f32(self._h, pid, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), n)
        else:
            buf = np.empty(n, np.int16)
            L.psk_harness_port_read_i16(self._h, pid, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_short)), n)
        return buf

    def packets(self, port):
        return _load().psk_harness_port_packets(self._h, PORTS[port])

    def sri_log(self, port):
        L = _load()
        x = (ctypes.c_double * 4096)()
        m = (ctypes.c_int * 4096)()
        n = L.psk_harness_port_sri(self._h, PORTS[port], x, m, 4096)
        return [(x[i], m[i]) for i in range(min(n, 4096))]

    @property
    def warnings(self):
        return _load().psk_harness_warnings(self._h)

    @property
    def last_eos(self):
        return bool(_load().psk_harness_last_eos(self._h))

    @property
    def last_stream(self):
        return _load().psk_harness_last_stream(self._h).decode()
