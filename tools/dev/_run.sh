cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4e
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "two_ranks or bench_two or clip_assembler" 2>&1 | tail -5
python bench.py > gpurun_out/r4e/bench_default.json 2> gpurun_out/r4e/bench_default.err; tail -3 gpurun_out/r4e/bench_default.err; python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r4e/bench_default.json") if l.startswith("{")][-1])
print({k:d[k] for k in ("value","ms_per_step")})
print("fp32", d.get("fps_fp32_convs"))
print("conv32", d.get("roofline_conv_fp32"))
print("conv", {k:d["roofline_conv"][k] for k in ("achieved","avg_us","frac_issued")})
print("ctx", d.get("context_error"))
PY
