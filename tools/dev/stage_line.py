"""One line per library variant: tile kernel and splat stage of the timed C3 clip (bench.py --no-extras)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-extras", "--no-cpu-baseline", "--steps", "3", "--warmup", "1"] + sys.argv[1:],
                   capture_output=True, text=True)
line = [l for l in r.stdout.splitlines() if l.startswith("{")]
if not line:
    print("FAILED", r.stderr[-500:]); sys.exit(1)
d = json.loads(line[-1]); ro = d["roofline"]
print(f"{os.path.basename(os.environ.get('SLR_SFS_AMD_LIB', 'default')):16s} fps {d['value']:7.2f} | tile us/frame avg {ro['avg_us']:6.1f} min {ro['min_us']:6.1f} frac {ro['frac']:.3f} frac_min {ro['frac_min_bytes']:.3f} | stage {ro['stage_us']:6.1f} ({ro['stage_frac']:.3f}) | parity {d['parity_err']['ok']}")
