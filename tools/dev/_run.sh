cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4g
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r4g/pytest_all.log 2>&1; tail -4 gpurun_out/r4g/pytest_all.log
python bench.py > gpurun_out/r4g/bench_default.json 2> gpurun_out/r4g/bench_default.err; tail -3 gpurun_out/r4g/bench_default.err; python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r4g/bench_default.json") if l.startswith("{")][-1])
print({k:d[k] for k in ("value","ms_per_step")}, "ctx", d.get("context_error"))
r=d["roofline"]; print("roofline", r["frac"], r["frac_min_bytes"], r["stage_frac"], r["avg_us"], r["stage_us"])
print("dropin", json.dumps(d.get("roofline_dropin"))[:1800])
PY
