#!/bin/bash
# round 3, GPU call 1: the matrix-core chain probe, the baseline of this box, the all-channels-at-zero-phase worst
# case, the staggered-start experiment, and when the waves of a launch start and end.  Output: gpurun_out/r03_p1/
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03_p1
mkdir -p $out
cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/chain_probe tools/micro/mfma_lds_chain_probe.hip > $out/chain_probe.txt 2>&1 \
  && timeout -k 10 120 /tmp/chain_probe >> $out/chain_probe.txt 2>&1
echo "chain probe done: $(tail -1 $out/chain_probe.txt | cut -c1-200)"
timeout -k 10 300 python bench.py > $out/bench_base.json 2> $out/bench_base.err || { echo "baseline bench failed"; tail -5 $out/bench_base.err; exit 1; }
python - <<PY
import json
d = json.load(open("$out/bench_base.json"))
print("baseline: ms_per_step %.4f launch_ms_avg %.4f frac %.4f chain_blocks %d check %s" % (d["ms_per_step"], d["roofline"]["launch_ms_avg"], d["roofline"]["frac"], d["kernel_stats"]["fit_chain_blocks"], d.get("check")))
PY
timeout -k 10 300 python bench.py --phase0 --no-cpu-baseline --no-few > $out/bench_phase0.json 2> $out/bench_phase0.err || { echo "phase0 bench failed"; tail -5 $out/bench_phase0.err; }
python - <<PY
import json
try:
    d = json.load(open("$out/bench_phase0.json"))
    print("phase0: ms_per_step %.4f launch_ms_avg %.4f frac %.4f chain_blocks %d of %d check %s" % (d["ms_per_step"], d["roofline"]["launch_ms_avg"], d["roofline"]["frac"], d["kernel_stats"]["fit_chain_blocks"], d["kernel_stats"]["unwrap_blocks"], d.get("check")))
except Exception as e:
    print("phase0: no line", e)
PY
cp psk_soft_amd/libpsk_soft_hip.so /tmp/lib_orig.so
timeout -k 10 900 tools/ab_bench_args.sh 3 "" "" "-DPSK_STAGGER=12" "-DPSK_STAGGER=24" "-DPSK_STAGGER=64" > $out/ab_stagger.txt 2>&1
grep VARIANT $out/ab_stagger.txt
# when the waves start and end (diagnostic build of the headline instantiation)
cd $R/psk_soft_amd/csrc && rm -f obj/psk_fast_S8_H1_E0.o && make -j16 EXTRA="-DPSK_DIAG_STAMP" > /tmp/make_diag.log 2>&1 || { echo "diag build failed"; tail -5 /tmp/make_diag.log; }
cd $R
timeout -k 10 300 python bench.py --steps 10 --warmup 10 --no-cpu-baseline --no-check --no-few --stamps $out/stamps_base.npy > $out/bench_diag.json 2> $out/bench_diag.err
timeout -k 10 300 python bench.py --phase0 --steps 10 --warmup 10 --no-cpu-baseline --no-check --no-few --stamps $out/stamps_phase0.npy > $out/bench_diag_phase0.json 2> $out/bench_diag_phase0.err
python - <<PY
import numpy as np
for name in ("base", "phase0"):
    try:
        a = np.load("$out/stamps_%s.npy" % name)
    except Exception as e:
        print(name, "no stamps", e); continue
    t0, t1, ch = a[:, 0], a[:, 1], a[:, 2]
    s0 = (t0 - t0.min()) % (1 << 32); e1 = (t1 - t0.min()) % (1 << 32)
    dur = (t1 - t0) % (1 << 32)
    q = lambda v, p: np.percentile(v, p) / 100.0
    print("%s: start spread us p50 %.1f p99 %.1f max %.1f | end us min %.1f p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f | duration us p50 %.1f max %.1f" % (
        name, q(s0, 50), q(s0, 99), q(s0, 100), q(e1, 0), q(e1, 10), q(e1, 50), q(e1, 90), q(e1, 99), q(e1, 100), q(dur, 50), q(dur, 100)))
    late = np.argsort(e1)[-12:]
    print("   last 12 waves: end us %s chained blocks %s" % (np.round(e1[late] / 100.0, 1).tolist(), ch[late].tolist()))
    print("   mean end of waves with chain blocks > 128: %.1f us (n=%d); of the others %.1f us" % (
        (e1[ch > 128].mean() / 100.0 if (ch > 128).any() else 0), int((ch > 128).sum()), e1[ch <= 128].mean() / 100.0))
PY
cp /tmp/lib_orig.so psk_soft_amd/libpsk_soft_hip.so
