#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): parity tests (both step kernels), bench, kernel trace and PMC passes.
# usage: tools/gpu_check.sh <tag> [notest] [nopmc]      outputs → gpurun_out/<tag>/
TAG=${1:-run}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
if [[ " $* " != *" notest "* ]]; then
  for K in ${TG_KERNELS:-bits block quad lanes wave}; do
    # (the earlier formulations live in the test-only library: make -C tetris_gymnasium_amd/csrc legacy)
    [[ $K == bits ]] && unset TG_LIB || export TG_LIB=$PWD/tetris_gymnasium_amd/libtetris_gym_hip_legacy.so
    TG_KERNEL=$K timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_$K.log 2>&1; echo "pytest[$K] rc=$?"; tail -3 $OUT/pytest_gpu_$K.log
  done
fi
for K in ${TG_KERNELS:-bits block quad lanes wave}; do
  [[ $K == bits ]] && unset TG_LIB || export TG_LIB=$PWD/tetris_gymnasium_amd/libtetris_gym_hip_legacy.so
  TG_KERNEL=$K timeout 600 python bench.py --steps 512 --warmup 64 --no-cpu-baseline > $OUT/bench_$K.json 2> $OUT/bench_$K.err; echo "bench[$K] rc=$?"; cut -c1-330 $OUT/bench_$K.json
done
if [[ " $* " != *" nopmc "* ]]; then
  KN=${TG_PROFILE_KERNEL:-step_kernel}
  rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python3 bench.py --steps 128 --warmup 16 --no-cpu-baseline > $OUT/kt.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d $OUT/pmc1 -o pmc1 -- python3 bench.py --steps 32 --warmup 8 --no-cpu-baseline > $OUT/pmc1.log 2>&1
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc2 -o pmc2 -- python3 bench.py --steps 32 --warmup 8 --no-cpu-baseline > $OUT/pmc2.log 2>&1
  for f in kt/kt pmc1/pmc1 pmc2/pmc2; do python3 profiles/summarize_rocprof.py $OUT/${f}_results.db $KN > $OUT/$(basename $f)_summary.md 2>&1; done
  rm -f $OUT/*/*.db
  grep -h "step_kernel" $OUT/kt_summary.md | head -3 | cut -c1-150; grep -h "SQ_" $OUT/pmc1_summary.md $OUT/pmc2_summary.md | cut -c1-120
fi
