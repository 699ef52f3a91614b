#!/bin/bash
# round 4, final profile set on the final kernel sources: bench lines of every config / dtype, rocprofv3 kernel stats (cfg2 / cfg4 / cfg5),
# PMC passes (FETCH_SIZE, WRITE_SIZE, MFMA busy; separate runs) for cfg2, cfg4, cfg5
cd "$(dirname "$0")/../.." && . tools/session/r4lib.sh
R=$PWD
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$R}
step r4z_bench_cfg2 600 python bench.py --details
grep '^{"metric"' gpurun_out/r4z_bench_cfg2.log | head -1 > gpurun_out/r4z_bench_cfg2.json
python - <<'PY'
t = open("gpurun_out/r4z_bench_cfg2.log").read()
i = t.find('{\n "conv_variants"')
if i >= 0:
    j = t.find("\n{\"metric\"", i)
    open("gpurun_out/r4z_conv_shapes.json", "w").write(t[i:j if j > 0 else None])
PY
step r4z_bench_cfg4 500 python bench.py --config cfg4 --steps 10 --warmup 2 --cpu-budget 30
step r4z_bench_cfg5 400 python bench.py --config cfg5 --steps 20 --warmup 2
step r4z_bench_cfg5_bf16 300 python bench.py --config cfg5 --dtype bf16 --steps 20 --warmup 2 --no-cpu-baseline --no-sample-call
step r4z_bench_cfg2_fp16 300 python bench.py --dtype fp16 --steps 50 --no-cpu-baseline --no-sample-call
step r4z_bench_cfg2_bf16 300 python bench.py --dtype bf16 --steps 50 --no-cpu-baseline --no-sample-call
step r4z_bench_cfg1 300 python bench.py --config cfg1 --steps 50 --cpu-budget 30
for c in cfg4 cfg5 cfg5_bf16 cfg2_fp16 cfg2_bf16 cfg1; do grep "^{" gpurun_out/r4z_bench_$c.log | head -1 > gpurun_out/r4z_bench_$c.json; done
cd /tmp && export TMPDIR=/tmp
prof() {   # prof <tag> <counters or ""> <bench args...>
    local tag=$1 ctr=$2; shift 2
    if [ -z "$ctr" ]; then
        ( timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag -- python3 $R/bench.py "$@" ) > $R/gpurun_out/$tag.log 2>&1
    else
        ( timeout -k 10 500 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $R/gpurun_out/$tag -- python3 $R/bench.py "$@" ) > $R/gpurun_out/$tag.log 2>&1
    fi
    local rc=$?; echo "$tag rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
}
Q="--no-cpu-baseline --no-roofline --no-sample-call --no-strict-f32"
prof r4z_stats "" --steps 10 --warmup 2 --no-cpu-baseline --no-sample-call --no-strict-f32 --no-live-traffic
prof r4z_fetch "FETCH_SIZE" --steps 2 --warmup 1 $Q
prof r4z_write "WRITE_SIZE" --steps 2 --warmup 1 $Q
prof r4z_mfma "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" --steps 2 --warmup 1 $Q
prof r4z4_stats "" --config cfg4 --steps 4 --warmup 1 --no-cpu-baseline --no-sample-call --no-strict-f32 --no-live-traffic
prof r4z4_fetch "FETCH_SIZE" --config cfg4 --steps 2 --warmup 1 $Q
prof r4z4_write "WRITE_SIZE" --config cfg4 --steps 2 --warmup 1 $Q
prof r4z5_stats "" --config cfg5 --steps 10 --warmup 2 --no-cpu-baseline --no-sample-call --no-live-traffic
prof r4z5_fetch "FETCH_SIZE" --config cfg5 --steps 2 --warmup 1 $Q
prof r4z5_write "WRITE_SIZE" --config cfg5 --steps 2 --warmup 1 $Q
prof r4z5_mfma "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" --config cfg5 --steps 2 --warmup 1 $Q
cd $R
# keep the merge small: counter / stats csv only
for d in gpurun_out/r4z_stats gpurun_out/r4z_fetch gpurun_out/r4z_write gpurun_out/r4z_mfma gpurun_out/r4z4_* gpurun_out/r4z5_*; do
    [ -d "$d" ] && find "$d" -type f ! -name "*counter_collection.csv" ! -name "*kernel_stats.csv" -delete 2>/dev/null
done
ls gpurun_out | grep r4z | head -40
