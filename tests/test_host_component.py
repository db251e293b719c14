"""The C++ host class (psk_soft_amd/host/psk_soft_gpu.h) behind in-memory ports: the part of
serviceFunction() that stays on the host -- packet ownership, the three pushSRI calls and
their rewritten xdelta/mode (reference cpp/psk_soft.cpp:399-404), push-only-if-non-empty and
T/EOS/streamID pass-through (:605-615), NOOP on an empty queue (:350-352), the real-data
warning (:359-363).  CPU part uses a control-plane-only component (no data); the GPU part
runs the reference's six component tests through it."""
import math

import numpy as np
import pytest

from psk_soft_amd import lib as pl
from psk_soft_amd import sandbox


def test_host_side_sri_and_packets_control_plane():
    comp = sandbox.Component(device=pl.DEVICE_NONE)
    comp.samplesPerBaud = 8
    comp.constelationSize = 8
    comp.numAvg = 100
    assert comp.service() == pl.NOOP  # nothing queued
    data = np.zeros(2 * 1000 * 8, np.float32)
    comp.push(data, sampleRate=100, sriChanged=True, EOS=False, streamID="abc", twsec=42.0)
    assert comp.service() == pl.NORMAL
    # SRI rewrite: soft xdelta*S, phase same with mode 0, bits additionally / bitsPerBaud
    (xs, ms), = comp.sri_log("softDecision_dataFloat_out")
    (xp, mp), = comp.sri_log("phase_dataFloat_out")
    (xb, mb), = comp.sri_log("bits_dataShort_out")
    assert xs == 0.01 * 8 and ms == 1
    assert xp == 0.01 * 8 and mp == 0
    assert xb == 0.01 * 8 / 3 and mb == 0
    assert comp.sri_log("sampleIndex_dataShort_out") == []  # the reference never pushes an SRI there
    assert comp.getData("softDecision_dataFloat_out").size == 2 * 901
    assert comp.getData("bits_dataShort_out").size == 3 * 901
    assert comp.getData("phase_dataFloat_out").size == 901
    assert comp.getData("sampleIndex_dataShort_out").size == 901
    assert comp.last_stream == "abc" and not comp.last_eos
    # a packet that completes no symbol pushes nothing, but (Q2) still pushes the three SRIs
    comp.push(np.zeros(6, np.float32), sampleRate=100, EOS=True)
    assert comp.service() == pl.NORMAL
    assert comp.packets("softDecision_dataFloat_out") == 1
    assert len(comp.sri_log("softDecision_dataFloat_out")) == 2
    # real data: warning, nothing pushed, no SRI
    comp.push(np.zeros(64, np.float32), sampleRate=100, complexData=False)
    assert comp.service() == pl.NORMAL
    assert comp.warnings == 1
    assert len(comp.sri_log("softDecision_dataFloat_out")) == 2
    # queue flush: warning + resetState handled inside the library, self-clearing
    comp.push(np.zeros(64, np.float32), sampleRate=100, inputQueueFlushed=True)
    assert comp.service() == pl.NORMAL
    assert comp.warnings == 2 and comp.resetState == 0
    comp.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["testDiffDecode8PSK", "testDiffDecodeBPSK", "testDiffDecodeQPSK",
                                  "testNonDiffDecode8PSK", "testNonDiffDecodeBPSK", "testNonDiffDecodeQPSK"])
def test_reference_component_tests_through_host_class(name):
    """reference tests/test_psk_soft.py:178-238, written the way the reference writes them."""
    from tests.test_oracle_reference_kat import reference_stimuli, to_cx

    numSyms, diff, data, syms = reference_stimuli()[name]
    comp = sandbox.Component(device=0)
    comp.samplesPerBaud = 8
    comp.constelationSize = numSyms
    comp.numAvg = 100
    comp.differentialDecoding = diff
    comp.push(data, sampleRate=100, complexData=True, sriChanged=True)
    assert comp.service() == pl.NORMAL
    outCx = to_cx(comp.getData("softDecision_dataFloat_out"))
    if diff:
        rot = complex(math.cos(math.pi / 4), math.sin(math.pi / 4)) if numSyms == 4 else 1
        maxError = max(abs(x - rot * y) for x, y in zip(outCx[1:], syms[1:]))
    else:
        thetas = {2: [0, math.pi], 4: [math.pi / 4 + k * math.pi / 2 for k in range(4)], 8: [k * math.pi / 4 for k in range(8)]}[numSyms]
        maxError = min(max(abs(complex(math.cos(t), math.sin(t)) * x - y) for x, y in zip(outCx[1:], syms[1:])) for t in thetas)
    assert maxError < 1e-3
    assert comp.getData("bits_dataShort_out").size == 901 * {2: 1, 4: 2, 8: 3}[numSyms]
    comp.close()
