#!/bin/bash
# Runs ON THE GPU BOX: the round's profile set → gpurun_out/<tag>/ (copy the summaries + latest.json into profiles/<round>/).
#   per block (config3 = headline, then the blocks of bench.py's block_specs under their profiles/latest.json names): rocprofv3
#   --kernel-trace --stats of the bench block's command
#   every block: HBM traffic (FETCH_SIZE, WRITE_SIZE in separate --pmc passes, calibrated on the stream-copy kernel)
#   config3: two SQ_* counter passes
# usage: tools/gpu_profile.sh <tag> [blocks...]
TAG=${1:-prof}; shift; BLOCKS=${@:-config3 config2 config5 dict fused16_traj rgb functional functional_fused16_traj config2_fused16_traj config2_dict config2_rgb}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
# (every block but the headline through --only-block: the arguments, and so the instantiation, of the default bench run)
declare -A ARGS=( [config3]="" [config2]="--only-block config2" [config5]="--only-block config5" [dict]="--only-block dict" [fused16_traj]="--only-block fused_rollout" [rgb]="--only-block rgb" [functional]="--only-block functional" [functional_fused16_traj]="--only-block functional_fused" [config2_fused16_traj]="--only-block config2_fused" [config2_dict]="--only-block config2_dict" [config2_rgb]="--only-block config2_rgb" [fused16]="--fuse 16" [functional_fused16]="--ruleset functional --obs fn --fuse 16" )
# (warm-up and steps are multiples of 16: every launch of a fused block then carries 16 transitions; --no-secondary keeps the
# tg_step_k comparison launches — another instantiation — out of the fused blocks' traces)
COMMON="--no-cpu-baseline --no-blocks --no-secondary --steps 128 --warmup 16"
for b in $BLOCKS; do
  rocprofv3 --kernel-trace --stats -d $OUT/${b}_kt -o kt -- python3 bench.py $COMMON ${ARGS[$b]} > $OUT/${b}_kt.log 2>&1
  python3 profiles/summarize_rocprof.py $OUT/${b}_kt/kt_results.db > $OUT/${b}_kt_summary.md 2>&1
  echo "== $b"; grep -h "step_kernel\|observe_kernel" $OUT/${b}_kt_summary.md | head -3 | cut -c1-140
done
for b in $BLOCKS; do
  case $b in *)
    rocprofv3 --pmc FETCH_SIZE -d $OUT/${b}_f -o f -- python3 bench.py $COMMON --steps 32 --warmup 16 ${ARGS[$b]} > $OUT/${b}_f.log 2>&1
    rocprofv3 --pmc WRITE_SIZE -d $OUT/${b}_w -o w -- python3 bench.py $COMMON --steps 32 --warmup 16 ${ARGS[$b]} > $OUT/${b}_w.log 2>&1
  ;; esac
done
# SQ counters (two passes) for the headline, and the instruction-mix pass for the functional and the fused blocks (what backs "the launch
# is bound by issue slots / by bytes" in DESIGN.md)
for b in $BLOCKS; do
  case $b in config3|functional|fused16_traj|functional_fused16_traj|config5)
    rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d $OUT/pmc1 -o pmc1 -- python3 bench.py $COMMON --steps 32 --warmup 16 ${ARGS[$b]} > $OUT/${b}_pmc1.log 2>&1
    python3 profiles/summarize_rocprof.py $OUT/pmc1/pmc1_results.db step_kernel > $OUT/${b}_pmc1_summary.md 2>&1
    if [[ $b == config3 ]]; then
      rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc2 -o pmc2 -- python3 bench.py $COMMON --steps 32 --warmup 16 > $OUT/pmc2.log 2>&1
      python3 profiles/summarize_rocprof.py $OUT/pmc2/pmc2_results.db step_kernel > $OUT/config3_pmc2_summary.md 2>&1
    fi
    rm -rf $OUT/pmc1 $OUT/pmc2
  ;; esac
done
python3 - "$OUT" $BLOCKS <<'PY'
import json, os, sqlite3, sys
out, blocks = sys.argv[1], sys.argv[2:]
def mean(db, kern, ctr):
    con = sqlite3.connect(db)
    rows = con.execute("select dispatch_id, sum(value) from counters_collection where kernel_name like ? and counter_name=? group by dispatch_id", (f"%{kern}%", ctr)).fetchall()
    v = [r[1] for r in rows]
    return (sum(v) / len(v), len(v)) if v else (0.0, 0)
def top(db, sub):
    con = sqlite3.connect(db)
    rows = con.execute("select name, total_calls, average from top_kernels").fetchall()
    rows = [r for r in rows if sub in r[0]]
    rows.sort(key=lambda r: -r[1] * r[2])
    if not rows: return None
    # a block may alternate between instantiations of the step kernel (launches with and without the RNG role's workgroups
    # pick different FEAT flags): the name of the one with the most time, calls and mean duration over all of them
    calls = sum(r[1] for r in rows)
    return (rows[0][0], calls, sum(r[1] * r[2] for r in rows) / calls, [(r[0].split("(")[0], r[1], r[2]) for r in rows])
latest = {}
for b in blocks:
    ent = {}
    db = f"{out}/{b}_kt/kt_results.db"
    if os.path.exists(db):
        main_sub = "step_kernel"
        t = top(db, main_sub)
        if t:
            ent.update(kernel=t[0], kernel_us=t[2] / 1e3 if t[2] > 1e3 else t[2], calls=t[1], kernel_source=f"profiles/ROUND/{b}_kt_summary.md")
            if len(t[3]) > 1: ent["instantiations"] = [{"kernel": n, "calls": c, "avg_us": (a / 1e3 if a > 1e3 else a)} for n, c, a in t[3]]
    fdb, wdb = f"{out}/{b}_f/f_results.db", f"{out}/{b}_w/w_results.db"
    if os.path.exists(fdb) and os.path.exists(wdb):
        tr = {}
        for kern in ("step_kernel", "stream_copy_kernel"):
            f, n1 = mean(fdb, kern, "FETCH_SIZE"); w, n2 = mean(wdb, kern, "WRITE_SIZE")
            tr[kern] = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "dispatches": n1}
        # stream copy of the same run: 1 GiB read + 1 GiB written per dispatch
        fc = tr["stream_copy_kernel"]["FETCH_SIZE_KB"] * 1024 / float(1 << 30) or 1.0
        wc = tr["stream_copy_kernel"]["WRITE_SIZE_KB"] * 1024 / float(1 << 30) or 1.0
        tr["calibration"] = {"fetch_reported_over_actual": fc, "write_reported_over_actual": wc}
        tr["hbm_read_bytes_per_launch"] = tr["step_kernel"]["FETCH_SIZE_KB"] * 1024 / fc
        tr["hbm_write_bytes_per_launch"] = tr["step_kernel"]["WRITE_SIZE_KB"] * 1024 / wc
        tr["hbm_bytes_per_launch"] = tr["hbm_read_bytes_per_launch"] + tr["hbm_write_bytes_per_launch"]
        tr["note"] = ("FETCH_SIZE / WRITE_SIZE (KB) from separate rocprofv3 --pmc passes, corrected by reported/actual measured on the "
                      "stream-copy kernel (1 GiB in, 1 GiB out) of the same run; these L2-side counters also count Infinity-Cache hits")
        json.dump(tr, open(f"{out}/{b}_traffic.json", "w"), indent=1)
        ent.update(hbm_bytes_per_launch=tr["hbm_bytes_per_launch"], traffic_source=f"profiles/ROUND/{b}_traffic.json")
    latest[b] = ent
json.dump(latest, open(f"{out}/latest.json", "w"), indent=1)
print(json.dumps(latest, indent=1))
PY
rm -rf $OUT/*_kt $OUT/*_f $OUT/*_w
