"""Generates tests/golden/psk_soft_golden.npz + manifest.json.

WHAT THESE FIXTURES ARE: inputs and the four output streams of the CPU oracle
(oracle/psk_soft_oracle.c, the build's restatement of reference cpp/psk_soft.cpp:346-618) on
the fixture list of SURVEY.md section 8(c).  The reference itself cannot be built or run in this
image (it needs the REDHAWK / BULKIO / boost headers), so these are NOT outputs of the reference;
they pin the oracle against drift and give the GPU parity tests answers that do not depend on
building the oracle on the GPU box.  The oracle in turn is pinned by the reference's own
known-answer tests (tests/test_oracle_reference_kat.py).

Run from the repository root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import pyoracle as po  # noqa: E402
from psk_soft_amd.stimulus import synth_channel  # noqa: E402

XDELTA = 0.01
HERE = os.path.dirname(os.path.abspath(__file__))


def packets(n, size):
    """[(start, stop)] in complex samples"""
    if not size:
        return [(0, n)]
    return [(p, min(p + size, n)) for p in range(0, n, size)]


def build_cases():
    cases = []
    # A: M x S x differential x packetisation, 4096 complex samples each
    for M in (2, 4, 8):
        for S in (8, 10):
            for diff in (0, 1):
                for pk in (0, 1000, 7):
                    cases.append(
                        dict(
                            name="a_m%d_s%d_d%d_p%d" % (M, S, diff, pk),
                            input=dict(channel=100 + M * 16 + S, M=M, S=S, n=4096),
                            props=dict(constelationSize=M, samplesPerBaud=S, differentialDecoding=diff),
                            events=[["packets", 0, 4096, pk]],  # the range cut into packets of pk samples (0: one)
                        )
                    )
    # B: cold start, then resetState before the third packet, then a flushed queue
    cases.append(
        dict(
            name="b_reset",
            input=dict(channel=200, M=4, S=8, n=8192),
            props=dict(constelationSize=4, samplesPerBaud=8),
            events=[["packet", 0, 2048], ["packet", 2048, 4096], ["set", "resetState", 1], ["packet", 4096, 6144],
                    ["packet_flushed", 6144, 8192]],
        )
    )
    # C: carrier offset large enough that the end-of-call wrap (cpp/psk_soft.cpp:592-603) runs several times
    cases.append(
        dict(
            name="c_wrap",
            input=dict(channel=201, M=4, S=8, n=16384, cfo_max=0.3, sigma=0.005),
            props=dict(constelationSize=4, samplesPerBaud=8),
            events=[["packets", 0, 16384, 2048]],
        )
    )
    # D: property changes mid-stream (phaseAvg and numAvg shrink / grow, samplesPerBaud change)
    cases.append(
        dict(
            name="d_props",
            input=dict(channel=202, M=4, S=8, n=12288),
            props=dict(constelationSize=4, samplesPerBaud=8),
            events=[["packet", 0, 2048], ["set", "phaseAvg", 20], ["packet", 2048, 4096], ["set", "numAvg", 40],
                    ["packet", 4096, 6144], ["set", "phaseAvg", 80], ["set", "numAvg", 120], ["packet", 6144, 8192],
                    ["set", "differentialDecoding", 1], ["packet", 8192, 10240], ["set", "constelationSize", 2],
                    ["packet", 10240, 12288]],
        )
    )
    return cases


# E: the 2^20-sample single-call runs of BASELINE configs[0] (BPSK) and configs[1] (QPSK): digests
# and the first / last 64 values of every stream (the input is regenerated from its seed)
LONG = [
    dict(name="e_c1_bpsk", input=dict(channel=300, M=2, S=8, n=1 << 20), props=dict(constelationSize=2, samplesPerBaud=8)),
    dict(name="e_c2_qpsk", input=dict(channel=301, M=4, S=8, n=1 << 20), props=dict(constelationSize=4, samplesPerBaud=8)),
]


def make_input(spec):
    kw = {k: spec[k] for k in ("cfo_max", "sigma") if k in spec}
    return synth_channel(spec["channel"], spec["M"], spec["S"], spec["n"], **kw)


def expand(events):
    """["packets", start, stop, size] -> one ["packet", a, b] per packet"""
    out = []
    for ev in events:
        if ev[0] == "packets":
            out += [["packet", ev[1] + a, ev[1] + b] for a, b in packets(ev[2] - ev[1], ev[3])]
        else:
            out.append(ev)
    return out


def run_case(case, iq):
    comp = po.OracleComponent()
    for k, v in case["props"].items():
        setattr(comp, k, v)
    out = dict(soft=[], bits=[], phase=[], index=[])
    first = True
    for ev in expand(case.get("events", [["packet", 0, iq.size // 2]])):
        if ev[0] == "set":
            setattr(comp, ev[1], ev[2])
            continue
        a, b = ev[1], ev[2]
        r = comp.service(iq[2 * a : 2 * b], XDELTA, sriChanged=first, inputQueueFlushed=(ev[0] == "packet_flushed"))
        first = False
        out["soft"].append(r.soft)
        out["bits"].append(r.bits)
        out["phase"].append(r.phase)
        out["index"].append(r.index)
    return {k: np.concatenate(v) for k, v in out.items()}


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    arrays, manifest = {}, dict(xdelta=XDELTA, cases=[], long=[])
    inputs = {}
    for case in build_cases():
        key = json.dumps(case["input"], sort_keys=True)
        if key not in inputs:
            inputs[key] = "in%d" % len(inputs)
            arrays[inputs[key]] = make_input(case["input"])
        iq = arrays[inputs[key]]
        out = run_case(case, iq)
        for k, v in out.items():
            arrays[case["name"] + "/" + k] = v
        manifest["cases"].append(dict(name=case["name"], input_key=inputs[key], props=case["props"], events=case["events"],
                                      n_symbols=int(out["phase"].size)))
    for case in LONG:
        iq = make_input(case["input"])
        out = run_case(case, iq)
        entry = dict(name=case["name"], input=case["input"], props=case["props"], input_sha256=digest(iq),
                     n_symbols=int(out["phase"].size), sha256={k: digest(v) for k, v in out.items()})
        for k, v in out.items():
            arrays[case["name"] + "/" + k + "_head"] = v[:64]
            arrays[case["name"] + "/" + k + "_tail"] = v[-64:]
        manifest["long"].append(entry)
    np.savez_compressed(os.path.join(HERE, "psk_soft_golden.npz"), **arrays)
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    print("cases:", len(manifest["cases"]), "long:", len(manifest["long"]), "arrays:", len(arrays))


if __name__ == "__main__":
    main()
