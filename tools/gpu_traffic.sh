#!/bin/bash
# Runs ON THE GPU BOX: HBM traffic counters (separate --pmc passes, as the MI355X guide prescribes) for the step
# kernel, calibrated on the stream-copy kernel of the same run (known bytes).  Output: gpurun_out/<tag>/traffic.json
TAG=${1:-traffic}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -d $OUT/f -o f -- python3 bench.py --steps 32 --warmup 8 --no-cpu-baseline --no-fused-rollout > $OUT/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/w -o w -- python3 bench.py --steps 32 --warmup 8 --no-cpu-baseline --no-fused-rollout > $OUT/w.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/h -o h -- python3 bench.py --steps 32 --warmup 8 --no-cpu-baseline --no-fused-rollout > $OUT/h.log 2>&1
python3 - <<PY
import sqlite3, json
def mean(db, kern, ctr):
    con = sqlite3.connect(db)
    rows = con.execute("select dispatch_id, sum(value) from counters_collection where kernel_name like ? and counter_name=? group by dispatch_id", (f"%{kern}%", ctr)).fetchall()
    v = [r[1] for r in rows]
    return sum(v) / max(1, len(v)), len(v)
out = {}
for kern in ("step_kernel", "stream_copy_kernel"):
    f, n1 = mean("$OUT/f/f_results.db", kern, "FETCH_SIZE")
    w, n2 = mean("$OUT/w/w_results.db", kern, "WRITE_SIZE")
    hit, _ = mean("$OUT/h/h_results.db", kern, "TCC_HIT_sum")
    miss, _ = mean("$OUT/h/h_results.db", kern, "TCC_MISS_sum")
    out[kern] = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "dispatches": n1, "TCC_HIT": hit, "TCC_MISS": miss}
copy_bytes = float(1 << 30)
sc = out["stream_copy_kernel"]
# calibration: what the counters report for a known 1 GiB read + 1 GiB write
out["calibration"] = {"fetch_reported_over_actual": sc["FETCH_SIZE_KB"] * 1024 / copy_bytes,
                      "write_reported_over_actual": sc["WRITE_SIZE_KB"] * 1024 / copy_bytes}
st = out["step_kernel"]
fc = out["calibration"]["fetch_reported_over_actual"] or 1.0
wc = out["calibration"]["write_reported_over_actual"] or 1.0
out["hbm_read_bytes_per_launch"] = st["FETCH_SIZE_KB"] * 1024 / fc
out["hbm_write_bytes_per_launch"] = st["WRITE_SIZE_KB"] * 1024 / wc
out["hbm_bytes_per_launch"] = out["hbm_read_bytes_per_launch"] + out["hbm_write_bytes_per_launch"]
out["l2_hit_rate_step_kernel"] = st["TCC_HIT"] / max(1.0, st["TCC_HIT"] + st["TCC_MISS"])
out["note"] = "FETCH_SIZE/WRITE_SIZE (KB) from separate rocprofv3 --pmc passes; corrected by the ratio reported/actual measured on the float4 stream-copy kernel (1 GiB in, 1 GiB out) of the same run; Infinity-Cache hits are counted by these L2-side counters"
json.dump(out, open("$OUT/traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $OUT/f $OUT/w $OUT/h
