#!/bin/bash
# XCD-aware tile order (tuning tile 108) against the product tile 11: time and FETCH_SIZE
cd /root/repo
for sh in 16,256,256,128,0,128,3,1 16,128,128,256,0,256,3,1; do
python tools/conv_bench.py --shape $sh --tiles 11,108 --gn --stats --check --rounds 7 --iters 10 2>&1 | grep -v amdgpu.ids
done
cd /tmp && export TMPDIR=/tmp
CDX_TUNE=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /root/repo/gpurun_out/r3v_fetch -- python3 /root/repo/tools/conv_bench.py --shape 16,256,256,128,0,128,3,1 --tiles 11,108 --gn --stats --rounds 1 --iters 3 > /root/repo/gpurun_out/r3v_fetch.log 2>&1
CDX_TUNE=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /root/repo/gpurun_out/r3v_write -- python3 /root/repo/tools/conv_bench.py --shape 16,256,256,128,0,128,3,1 --tiles 11,108 --gn --stats --rounds 1 --iters 3 > /root/repo/gpurun_out/r3v_write.log 2>&1
echo rc=$?
