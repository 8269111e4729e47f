#!/bin/bash
# Runs ON THE GPU BOX: the host-facing surfaces (adapter, single env, device-resident) + their bench blocks (round 4).
TAG=${1:-host}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_surfaces.py tests/test_gpu_device_resident.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "== pytest surfaces + device-resident"; tail -15 $OUT/pytest.log
timeout 600 python - > $OUT/blocks.json 2> $OUT/blocks.err <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
out = {}
for name, fn in (("vector_env", bench.vector_env_block), ("single_env", bench.single_env_block), ("vector_env_shards", bench.shards_block)):
    try:
        out[name] = fn(0)
    except Exception as ex:
        import traceback; traceback.print_exc()
        out[name] = {"error": repr(ex)}
print(json.dumps(out, indent=1))
PY
echo "== bench blocks"; cat $OUT/blocks.json | head -150; tail -20 $OUT/blocks.err
