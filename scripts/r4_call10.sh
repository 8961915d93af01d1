mkdir -p gpurun_out/r4c10
timeout 500 python scripts/gpu_odometry_profile.py --frames 100 > gpurun_out/r4c10/odo_profile.json 2>gpurun_out/r4c10/err.txt
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4c10/odo_profile.json"))
for k,v in sorted(d.items(), key=lambda kv: -(kv[1]["total_ms"] if isinstance(kv[1],dict) else 0)):
    print(k, v)
PY
tail -3 gpurun_out/r4c10/err.txt
