#!/usr/bin/env python
"""bench.py -- headline benchmark of BASELINE.json on MI355X.

Metric  : synthesised frames/sec at 768x1280, N=60 (whole job, all GPUs)
Workload: config C3 of BASELINE.json -- the full baseline pipeline
          image -> encoder -> [per frame: Euler displacement maps -> fused softmax-splat of
          64 features (both directions) + normalisation -> partial-conv decoder -> tanh]
          on a synthetic 768x1280 image + smooth motion field, random-init weights of the
          reference architecture, fp32.  (--workload c4: the 2-layer SLR v1 pipeline.)
A step  : ONE 60-frame clip, everything included (both all-frames Euler passes, row lists + plan of
          all 120 displacement maps, encoder, 60 x (fused splat + decoder)), frames sharded round-robin
          over the ranks and assembled with ONE RCCL all-gather of the finished clip (the form north_star
          names: --assembly final --encoder redundant, the default) -> "scaling": "strong" (total work fixed).
          N > 1 also reports `value_rounds_banded`: the same clip with one hidden all-gather per round of frames
          and the encoder split into row bands (--assembly rounds --encoder banded), measured after the timed steps.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 3 --warmup 1

Prints ONE JSON line on rank 0 (contract in the task statement) with extra objects:
  roofline     : the fused clip kernel (slr::clip_tile_kernel<G2,false,B4 = true>, csrc/splat_clip.hip; its feature planes plane-blocked by 4: the kernel that does
                 the exp-weighted two-direction splat + normalisation of one frame), timed
                 with HIP events on its launch stream inside the timed steps.
                 algorithmic bytes per launch = 2 * (2*65+2)*H*W*4 = 1038.1 MB at C3
                 (B_sum per reference splat call x the 2 calls of one frame).
                 stage_us / stage_frac: the WHOLE splat stage per frame inside the same steps -- the tile kernel, the
                 (normally empty) pass-by-pass launch, and the per-clip motion work (Euler passes, row lists, plan)
                 divided by the frames.  traffic: static_traffic() -- PMC passes kept under profiles/.
  parity_err   : decoder input of one frame of the TIMED clip against the CPU oracle, and the fused kernel on the
                 seeded 768x1280 inputs against digests of the reference's own forward_flow (outside the timed region).
  cpu_baseline : the CPU oracle (oracle/, OpenMP over planes) on this box's host cores, same
                 hot path (Euler + 2 x 65-plane splat + normalise) on a sample of frames; plus one thread.
  roofline_dropin / roofline_backward / roofline_conv / roofline_conv_fp32 / c4 / fps_fp32_convs : context measured after the timed
                 region (N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, W, NFRAMES = 768, 1280, 60
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
TRAFFIC_ROUND = "r6"            # profiles/<round>_traffic_<name>.json: PMC passes (tools/collect_profiles.sh -> tools/pmc_kernel_traffic.py) of one
                                # kernel each; every file carries the hash of the kernel sources it was taken on


def static_traffic(name, units):
    """(HBM-side bytes of one launch doing `units` units of work, where the figure comes from) out of profiles/rN_traffic_<name>.json --
    FETCH_SIZE x 2 + WRITE_SIZE per dispatch of that kernel, measured under rocprofv3 --pmc in separate passes, NOT in this run; a
    figure taken on other kernel sources is labelled stale.  (None, None) without the file."""
    rel = f"profiles/{TRAFFIC_ROUND}_traffic_{name}.json"
    tf = os.path.join(ROOT, rel)
    if not os.path.exists(tf):
        return None, None
    tj = json.load(open(tf))
    if "traffic_bytes_per_unit" not in tj:
        return None, None
    same = tj.get("source_sha16") == csrc_hash()
    src = (f"static: PMC passes (FETCH_SIZE x2 -- calibrated on this access pattern, profiles/r5_fetch_calibration.txt -- / WRITE_SIZE) of {rel} per unit of work x the units of this launch, not measured in "
           "this run; " + ("taken on these kernel sources" if same else
                           "STALE: taken on other kernel sources (re-run tools/collect_profiles.sh)"))
    return round(tj["traffic_bytes_per_unit"] * units), src


def measured_traffic(workload):
    """HBM-side bytes of the dominant kernel (the fused clip kernel) per frame of work, MEASURED IN THIS RUN: after the timed steps, two child
    processes run the splat stage of the same workload (tools/splat_stage.py: same sizes, same kernels, synthetic feature planes) under
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `... --pmc WRITE_SIZE` (separate passes, kernel-trace only, as MI355X_MICROARCH.md
    prescribes); FETCH_SIZE x 2 (gfx950 tallies 128-byte requests at 64; calibrated on this access pattern, profiles/r5_fetch_calibration.txt),
    WRITE_SIZE exact, both in KiB, mean over the kernel's dispatches / the 15 frames a dispatch of that command renders on average.
    -> (bytes per frame, description) or (None, reason)."""
    import csv, glob, shutil, subprocess, tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not on this box"
    pat = "clip_tile_kernel<false, false, true>" if workload == "c3" else "clip_tile_kernel<true, false, true>"
    tmp = tempfile.mkdtemp(prefix="slr_pmc_", dir="/tmp")
    got = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", os.path.join(tmp, counter), "-o", "p", "--",
                   sys.executable, os.path.join(ROOT, "tools", "splat_stage.py")] + (["v1"] if workload != "c3" else [])
            try:
                subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, TMPDIR="/tmp"), timeout=170, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            except Exception as e:
                return None, f"rocprofv3 pass {counter} failed: {type(e).__name__}"
            vals = []
            for f in glob.glob(os.path.join(tmp, counter, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if pat in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                        vals.append(float(row["Counter_Value"]))
            if not vals:
                return None, f"rocprofv3 pass {counter}: no dispatch of {pat} in the counter file"
            got[counter] = (sum(vals) / len(vals), len(vals))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    per_dispatch = got["FETCH_SIZE"][0] * 1024 * 2.0 + got["WRITE_SIZE"][0] * 1024
    return per_dispatch / 15.0, (f"measured in this run: rocprofv3 --pmc FETCH_SIZE (x2) / WRITE_SIZE, separate passes over tools/splat_stage.py after the "
                                 f"timed steps, {got['FETCH_SIZE'][1]} + {got['WRITE_SIZE'][1]} dispatches of {pat}; read "
                                 f"{got['FETCH_SIZE'][0] * 2048 / 15e6:.1f} MB + written {got['WRITE_SIZE'][0] * 1024 / 15e6:.1f} MB per frame")


def csrc_hash():
    """sha256 (first 16 hex digits) over the sources of the splat kernels (the files the fused tile kernel is built from)."""
    import hashlib
    h = hashlib.sha256()
    for name in ("slr_common.hpp", "slr_tuning.hpp", "splat_types.hpp", "splat_core.hpp", "splat_tile.hpp", "splat_rows.hpp", "splat_ws.hpp",
                 "splat_clip.hip", "splat_op.hip", "grad.hip"):
        h.update(name.encode())
        h.update(open(os.path.join(ROOT, "slr-sfs_amd", "csrc", name), "rb").read())
    return h.hexdigest()[:16]


def smooth_motion(h, w, seed=0, amp=1.5):
    """sinusoidal field, A = 1.5 px/frame, static left 35 %."""
    rng = np.random.default_rng(seed)
    p1, p2 = rng.uniform(0, 2 * np.pi, 2)
    y, x = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
    u = amp * np.sin(2 * np.pi * (2 * x / w + y / h) + p1)
    v = amp * np.cos(2 * np.pi * (x / w - 1.5 * y / h) + p2)
    m = (x >= 0.35 * w).astype(np.float32)
    return np.stack([u * m, v * m])[None].astype(np.float32)


def splat_alg_bytes(c_splat):
    """B_sum = (2C+2)*H*W*4 per reference splat call, two calls (directions) per frame."""
    return 2 * (2 * c_splat + 2) * H * W * 4


def make_step(model, image, motion, rank, world, assembly, encoder, frames_u8=False):
    from slr_sfs_amd import parallel
    mine = parallel.shard_frames(NFRAMES, rank, world)
    shard = (rank, world) if (world > 1 and encoder == "banded") else None

    def step():
        if world == 1:
            return model.synthesize(image, motion, NFRAMES, frames=mine)
        if assembly == "final":          # north_star form: one all-gather of the finished clip
            local = model.synthesize(image, motion, NFRAMES, frames=mine, shard=shard)
            if frames_u8:                # every rank converts its own frames; the collective moves 1 byte per sample
                local = parallel.frames_for_assembly(local)
            return parallel.gather_clip(local, NFRAMES, rank, world)
        # one small asynchronous all-gather per round of `world` frames, under the next round's rendering
        asm = parallel.ClipAssembler(NFRAMES, rank, world)
        model.synthesize(image, motion, NFRAMES, frames=mine, on_frame=asm.push, shard=shard)
        return asm.finish(like=image[0])
    return step


def timed_clips(step, steps, warmup, world, dev):
    """W untimed + K timed clips, barrier + synchronize on both sides, max over ranks.
    -> (seconds, last clip, tile-kernel events, stage events)"""
    from slr_sfs_amd import synthesis

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    synthesis.kernel_timing, synthesis.stage_timing = [], []
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        clip = step()
    fence()
    dt = time.perf_counter() - t0
    kev, synthesis.kernel_timing = synthesis.kernel_timing, None
    sev, synthesis.stage_timing = synthesis.stage_timing, None
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    return dt, clip, kev, sev


def splat_roofline(kev, sev, c_splat, kernel, measured=None):
    """Tile kernel (HIP events recorded by the library around that launch) and the whole stage per frame."""
    # one launch of the tile kernel does the work of `nf` frames (pipeline.DECODE_BATCH of them share a launch)
    launches = [(e0.elapsed_time(e1) * 1e3, nf) for e0, e1, nf in kev]
    per_frame = sorted(us / nf for us, nf in launches)
    nframes = sum(nf for _, nf in launches)
    k_avg = sum(us for us, _ in launches) / nframes                     # launch duration per frame of work
    l_avg = sum(us for us, _ in launches) / len(launches)
    fpl = nframes / len(launches)
    alg = splat_alg_bytes(c_splat)
    ach = alg / (k_avg * 1e-6) / 1e9
    frames = [(e0.elapsed_time(e1) * 1e3, nf) for k, e0, e1, nf in sev if k == "frame"]
    prep = [e0.elapsed_time(e1) * 1e3 for k, e0, e1, nf in sev if k in ("prep", "prep+")]
    nclips = sum(1 for k, *_ in sev if k == "prep")
    stage_us = (sum(us for us, _ in frames) + sum(prep)) / max(1, sum(nf for _, nf in frames))
    # the fused operator's own minimum traffic: 64 feature planes + Z + 2 x 2 displacement planes in, 64 planes out
    min_bytes = (c_splat - 1 + 1 + 4 + c_splat - 1) * H * W * 4
    traffic, src = static_traffic("clip_c3" if c_splat == 65 else "clip_c4", fpl)
    if measured is not None and measured[0]:              # bytes per frame of work from this run's own counter passes
        traffic, src = round(measured[0] * fpl), measured[1]
    elif measured is not None and src:
        src += f" (in-run counter passes unavailable: {measured[1]})"
    return {"bound": "hbm", "kernel": kernel, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": src,
            "frac_traffic": None if not traffic else round(traffic / (l_avg * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
            "alg_bytes_per_launch": round(alg * fpl), "frames_per_launch": round(fpl, 2), "launch_avg_us": round(l_avg, 1),
            "alg_bytes_per_frame": alg, "min_bytes_per_frame": min_bytes,
            "frac_min_bytes": round(min_bytes / (k_avg * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "avg_us": round(k_avg, 1), "min_us": round(per_frame[0], 1),
            "max_us": round(per_frame[-1], 1), "launches": len(launches),
            "note": "avg/min/max_us = launch duration / frames in the launch; achieved = alg_bytes_per_launch / launch_avg_us",
            "stage_us": round(stage_us, 1), "stage_frac": round(alg / (stage_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
            "stage_prep_us_per_clip": round(sum(prep) / max(1, nclips), 1),
            "stage": "per frame: fused tile kernel + the (normally empty) pass-by-pass launch; per clip / frames: both "
                     "all-frames Euler passes + row lists and plan of all displacement maps + the feature planes packed by 4"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3", choices=["c3", "c4"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the context measurements after the timed region")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 counter passes (FETCH_SIZE / WRITE_SIZE of the clip kernel) after the timed steps")
    ap.add_argument("--full-line", action="store_true", help="also put the context objects into the last line (rounds 1-5 format, > 10 KB)")
    ap.add_argument("--assembly", default="final", choices=["rounds", "final"],
                    help="N>1: ONE all-gather of the finished clip (default, the north_star form) | all-gather per round of frames "
                         "under the next round")
    ap.add_argument("--frames", default="fp32", choices=["fp32", "uint8"],
                    help="N > 1, --assembly final: what the all-gather moves -- the fp32 frames (11.8 MB each), or the uint8 frames every "
                         "rank made of its own (*0.5+0.5, *255, rounded: what the reference's writer saves; 2.95 MB each)")
    ap.add_argument("--encoder", default="redundant", choices=["banded", "redundant"],
                    help="N>1: every rank encodes the image (default) | per-clip encoder in row bands + one all-gather")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    # development only (tests/test_gpu_parity.py::test_bench_two_ranks_on_one_gpu): all ranks on cuda:0 with the
    # collectives over gloo, to exercise this script's multi-rank control flow on a one-GPU box
    one_gpu = os.environ.get("SLR_BENCH_ONE_GPU_GLOO") == "1"
    if one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import slr_sfs_amd as S
    from slr_sfs_amd import parallel, pipeline
    S._lib.lib()                                   # fail loudly if the HIP library is missing

    torch.manual_seed(0)
    model = (pipeline.BaselineAnimator() if a.workload == "c3" else pipeline.SLRv1Animator()).to(dev).eval()
    rng = np.random.default_rng(0)
    image = torch.from_numpy(rng.uniform(-1, 1, (1, 3, H, W)).astype(np.float32)).to(dev)
    motion = torch.from_numpy(smooth_motion(H, W)).to(dev)

    u8 = world > 1 and a.frames == "uint8" and a.assembly == "final"
    step = make_step(model, image, motion, rank, world, a.assembly, a.encoder, frames_u8=u8)
    dt, clip, kev, sev = timed_clips(step, a.steps, a.warmup, world, dev)
    assert clip.shape == ((NFRAMES, H, W, 3) if u8 else (NFRAMES, 3, H, W)) and (u8 or bool(torch.isfinite(clip).all()))

    c_splat = 65 if a.workload == "c3" else 67              # planes per reference splat call (v1: 67)
    extra, cpu, parity = {}, None, None
    if world > 1 and (a.assembly, a.encoder) == ("final", "redundant"):
        # context beside the contract form: per-round hidden all-gathers + banded encoder (every rank takes part)
        step2 = make_step(model, image, motion, rank, world, "rounds", "banded")
        dt2, clip2, _, _ = timed_clips(step2, max(1, min(a.steps, 3)), 1, world, dev)
        extra["value_rounds_banded"] = {"value": round(NFRAMES * max(1, min(a.steps, 3)) / dt2, 3), "unit": "frames/s",
                                        "form": "all-gather per round of frames under the next round + encoder in row bands"}
        del clip2
    if world > 1:
        # what the communicator says about this job: every rank reports its device and its own rate
        my_frames = len(parallel.shard_frames(NFRAMES, rank, world)) * a.steps
        extra["communicator"] = parallel.communicator_report(dev, my_frames, dt)
        assert extra["communicator"]["world_size"] == world == dist.get_world_size()
    if rank == 0 and world == 1:
        parity = parity_check(model, image, motion, a.workload, dev)
    if rank == 0 and world == 1 and not a.no_extras:
        try:                                             # context legs never cost the contract line
            extra.update(context_measurements(a.workload, image, motion, dev))
        except Exception as e:
            extra["context_error"] = f"{type(e).__name__}: {e}"[:500]
        if not a.no_cpu_baseline:
            cpu = cpu_baseline(motion.cpu().numpy())
    # HBM bytes of the dominant kernel from counter passes taken NOW, on this box (child processes; the static file is the fallback)
    meas = None
    if rank == 0 and world == 1 and not a.no_pmc and not a.no_extras:
        try:
            meas = measured_traffic(a.workload)
        except Exception as e:
            meas = (None, f"{type(e).__name__}: {e}"[:200])
    roofline = splat_roofline(kev, sev, c_splat, "slr::clip_tile_kernel<false,false,true>" if a.workload == "c3" else "slr::clip_tile_kernel<true,false,true>", meas)

    if rank == 0:
        frame_bytes = 3 * H * W * 4
        mine = len(parallel.shard_frames(NFRAMES, 0, world))
        if world == 1:
            par = "1 GPU, no collective"
            moved = 0
        else:
            enc = ("encoder in row bands + one all-gather of the 65 feature planes" if a.encoder == "banded"
                   else "every rank runs the encoder")
            asm = (f"all-gather per round of {world} frames under the next round" if a.assembly == "rounds"
                   else "ONE all-gather of the finished clip (north_star form)")
            par = f"frames t = r mod {world} per rank; {asm}; {enc}"
            rounds = parallel.frames_per_rank(NFRAMES, world)
            if u8:
                frame_bytes //= 4
                asm += ", uint8 frames converted per rank"
                par = f"frames t = r mod {world} per rank; {asm}; {enc}"
            moved = rounds * frame_bytes * (world - 1)              # received per rank for the clip assembly
            if a.encoder == "banded":
                moved += 65 * H * W * 4 * (world - 1) // world      # + the other ranks' encoder bands
        line = {
            "metric": "synthesised frames/sec at 768x1280 N=60",
            "value": round(NFRAMES * a.steps / dt, 3), "unit": "frames/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None,
            "dtype": "f32 (splat path: fp32 throughout; encoder/decoder convolutions: fp32 in/out and accumulation, operands as "
                     "2 x f16 splits = 22 significant bits, 3 MFMAs per product -- NOT plain fp32 multiplies; the all-fp32 figure "
                     "is fps_fp32_convs)",
            "data": "synthetic",
            "config": {"workload": ("C3 baseline pipeline encoder->Euler->softmax-splat->pconv2 decoder"
                                    if a.workload == "c3" else
                                    "C4 SLR-v1 2-layer pipeline (fluid + background + alpha)") +
                                   ", 768x1280, N=60, random-init weights of the reference architecture",
                       "frames_per_step": NFRAMES, "H": H, "W": W, "parallelism": par,
                       "assembly": a.assembly if world > 1 else None, "encoder": a.encoder if world > 1 else None,
                       "assembled_frames": ("uint8" if u8 else "fp32") if world > 1 else None,
                       "frames_rank0": mine, "collective_bytes_received_per_rank_per_clip": moved},
            "roofline": roofline, "parity_err": parity, "cpu_baseline": cpu,
        }
        emit(line, extra, a.full_line)
    if world > 1:
        dist.barrier()                               # rank 0's extra measurements are done: leave together
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------ output

def _pick(d, *keys):
    return None if not d else {k: d[k] for k in keys if k in d}


def emit(line, extra, full=False):
    """Context first, contract last.  Every context object goes out on a line of its own, prefixed `#ctx <name> ` (not a JSON line: a reader
    that takes the first or the last JSON line of stdout finds the contract line either way); the LAST line is the contract of the task
    statement, compact (about 2 KB) so that a record keeping the tail of stdout holds it whole: metric, value, config, dtype, roofline,
    cpu_baseline, parity_err and the headline figure of every context leg -- the strict-fp32 clip rate (`fps_fp32_convs`), the other
    pipeline (`c4_fps` / `c3_fps`), the drop-in operator, the training shape, the backward."""
    ctx = {"roofline_full": line["roofline"], "parity_err_full": line["parity_err"], "cpu_baseline_full": line["cpu_baseline"]}
    ctx.update(extra)
    for k, v in ctx.items():
        if v is not None:
            print(f"#ctx {k} " + json.dumps(v), flush=True)
    if full:
        big = dict(line)
        big.update(extra)
        print("#ctx full_line " + json.dumps(big), flush=True)
    c = dict(line)
    c["dtype"] = "f32 (convs: fp32 in/out/accumulate, operands 2 x f16 splits = 22 bits; all-fp32: fps_fp32_convs)"
    c["config"] = {k: v for k, v in line["config"].items() if v is not None and k not in ("H", "W", "frames_per_step")}
    c["config"]["workload"] = c["config"]["workload"].replace(", random-init weights of the reference architecture", ", random weights").replace(
        " encoder->Euler->softmax-splat->pconv2 decoder", "")
    c["config"].pop("frames_rank0", None)
    r = line["roofline"]
    c["roofline"] = _pick(r, "bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "frac_traffic", "alg_bytes_per_launch",
                          "launch_avg_us", "frac_min_bytes", "avg_us", "stage_us", "stage_frac", "stage_prep_us_per_clip")
    src = r.get("traffic_source") or ""
    c["roofline"]["traffic_source"] = "measured in this run" if src.startswith("measured in this run") else ("static file" if src else None)
    pe = line["parity_err"]
    if pe:
        c["parity_err"] = {"ok": pe.get("ok"), "tolerance": pe.get("tolerance"),
                           "decoder_input_vs_oracle": pe.get("timed_clip_decoder_input_vs_oracle_max_abs"),
                           "vs_reference_forward_flow_digest": pe.get("reference_forward_flow_digest_max_abs"),
                           "frames_vs_reference_models": pe.get("frames_vs_reference_models_max_abs"),
                           "frames_256_vs_reference_models": pe.get("frames_256_vs_reference_models_digest_max_abs"),
                           "holes": pe.get("holes"), "reference_holes": pe.get("reference_holes")}
        c["parity_err"] = {k: (float(f"{v:.3g}") if isinstance(v, float) else v) for k, v in c["parity_err"].items()}
    cb = line["cpu_baseline"]
    if cb:
        c["cpu_baseline"] = {"value": cb["value"], "unit": "frames/s (splat stage)", "cores": cb["cores"], "kind": cb["kind"],
                             "sample": cb["sample"].replace(" of the same 768x1280 N=60 clip", ""),
                             "one_thread": cb["one_thread"]["value"], "with_decoder": cb["with_decoder"]["value"]}
    f32 = extra.get("fps_fp32_convs")
    if f32:
        c["fps_fp32_convs"] = f32["value"]                       # the same clip with every convolution in plain fp32 (direct kernels)
        c["fps_fp32_winograd"] = f32["winograd"]["value"]
    for other in ("c3", "c4"):
        if other in extra:
            c[f"{other}_fps"] = extra[other]["value"]
            c[f"{other}_roofline_frac"] = extra[other]["roofline"]["frac"]
            c[f"{other}_parity_ok"] = (extra[other].get("parity_err") or {}).get("ok")
    d = extra.get("roofline_dropin")
    if d:
        fl = d["flows"]
        c["dropin"] = {"is": "one-flow call: [us, frac of 8 TB/s]", "frac": d.get("frac")}
        for k in ("euler_t30", "euler_t59", "identity", "incoherent"):
            if k in fl:
                c["dropin"][k] = [fl[k]["call_us"], fl[k]["call_frac"]]
        c["dropin"]["c2"] = [d["c2"]["call_us"], d["c2"]["call_frac"]]
        for k, v in d.get("small_grids", {}).items():
            c["dropin"][k] = [v["call_us"], v["call_frac"]]
        if "c2_batched" in d:
            c["dropin"]["c2_batched_per_sample"] = [d["c2_batched"]["per_sample_us"], d["c2_batched"]["call_frac"]]
        if "train_shape" in d:
            c["train_shape"] = {k: [v["fwd_us"], v["bwd_us"], v["frac"]] for k, v in d["train_shape"]["flows"].items()}
            c["train_shape"]["is"] = "[2,65,256,256]: [fwd us, bwd us, frac]"
    b = extra.get("roofline_backward")
    if b:
        c["backward"] = {k: [v["avg_us"], v["frac"]] for k, v in b["flows"].items()}
    for k in ("splat_stage_fps_1gpu", "context_error"):
        if k in extra:
            c[k] = extra[k]
    if "value_rounds_banded" in extra:
        c["value_rounds_banded"] = _pick(extra["value_rounds_banded"], "value", "unit")
    if "communicator" in extra:                                 # the proof of which ranks / devices took part stays in the contract line
        cm = extra["communicator"]
        c["communicator"] = _pick(cm, "backend", "rccl_version", "world_size", "distinct_devices")
        c["communicator"]["ranks"] = [_pick(r, "rank", "device_index", "device_name", "pci_bus_id", "frames", "fps_this_rank") for r in cm["ranks"]]
    c = {k: v for k, v in c.items() if v is not None or k in ("vs_baseline", "cpu_baseline", "parity_err")}      # (contract keys stay, null or not)
    c["context"] = "#ctx lines above"
    print(json.dumps(c), flush=True)


# ------------------------------------------------------------------------------------------------ parity

def a6_large_inputs(h=768, w=1280):
    """Seeded inputs of tests/golden/pipeline_a6_large.npz (same generator as tools/make_golden_pipeline.py and
    tests/conftest.py): only digests of the REFERENCE's forward_flow outputs are stored."""
    rng = np.random.default_rng(2000 + h)
    y, x = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
    u = 1.5 * np.sin(2 * np.pi * (2 * x / w + y / h) + 0.9)
    v = 1.5 * np.cos(2 * np.pi * (x / w - 1.5 * y / h) + 0.4)
    m = (x >= 0.35 * w).astype(np.float32)
    motion = np.stack([u * m, v * m])[None].astype(np.float32)
    fs = rng.standard_normal((1, 64, h, w)).astype(np.float32)
    Z = rng.standard_normal((1, 1, h, w)).astype(np.float32)
    alpha_out = rng.standard_normal((1, 2, h, w)).astype(np.float32)
    return fs, Z, motion, alpha_out


@torch.no_grad()
def parity_check(model, image, motion, workload, dev, t=30):
    """Outside the timed region.  (1) The decoder input of frame t of the clip that was just timed (same model, image,
    motion) against the CPU oracle on the same encoder outputs.  (2) The same fused kernel on the seeded inputs of
    tests/golden/pipeline_a6_large.npz against digests of the reference's own forward_flow at this grid."""
    from oracle import oracle as o           # test infrastructure, used here only as the checker
    from slr_sfs_amd import synthesis
    o.build()
    out = {"t": t, "tolerance": 1e-4}
    clip = model.begin_clip(image, motion, NFRAMES, frames=[t])
    feats = clip.features(t)
    mnp = motion.cpu().numpy()
    if workload == "c3":
        ref = o.synth_baseline(clip.fs.cpu().numpy(), clip.Z.cpu().numpy(), mnp, t, NFRAMES)
        err = float(np.abs(feats.cpu().numpy() - ref).max())
    else:
        gen, afl = feats
        rg, ra, _ = o.synth_v1(clip.fs.cpu().numpy(), clip.Z.cpu().numpy(), clip.af.cpu().numpy(),
                               clip.alpha_bg.cpu().numpy(), mnp, t, NFRAMES)
        err = max(float(np.abs(gen.cpu().numpy() - rg).max()), float(np.abs(afl.cpu().numpy() - ra).max()))
    out["timed_clip_decoder_input_vs_oracle_max_abs"] = err
    gpath = os.path.join(ROOT, "tests", "golden", "pipeline_a6_large.npz")
    if os.path.exists(gpath):
        g = np.load(gpath)
        fs, Z, mo, a = a6_large_inputs(H, W)
        d = lambda x: torch.from_numpy(x).to(dev)
        kind = "baseline" if workload == "c3" else "v1"
        if workload == "c3":
            gen = synthesis.ClipSynthesizer(d(fs), d(Z), d(mo), NFRAMES, frames=[t]).features(t)
        else:
            gen, _ = synthesis.ClipSynthesizer(d(fs), d(Z), d(mo), NFRAMES, alpha_fluid_logit=d(a[:, 1:2]),
                                               alpha_bg=torch.sigmoid(d(a[:, 0:1])), frames=[t]).features(t)
        gen = gen.cpu().numpy()
        out["reference_forward_flow_digest_max_abs"] = float(np.abs(gen.ravel()[g["c3_pos"]] - g[f"c3_{kind}_t{t}_val"]).max())
        out["reference_holes"] = int(g[f"c3_{kind}_t{t}_holes"])
        out["holes"] = int((gen == 0).sum())
    # (3) FRAMES: the whole pipeline (encoder -> Euler -> splat -> decoder, HIP kernels throughout) on the seeded weights /
    # image / motion of tests/golden/pipeline_e2e.npz against the frames the reference's own models produced from them
    # (tools/make_golden_e2e.py; 64x64, N = 8) -- the north star's parity statement on the frames themselves.
    epath = os.path.join(ROOT, "tests", "golden", "pipeline_e2e.npz")
    if os.path.exists(epath):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import nets_fixture as NF
        from slr_sfs_amd import nets, pipeline
        g, gn = np.load(epath), np.load(os.path.join(ROOT, "tests", "golden", "nets_reference.npz"))
        img, mo, n = NF.e2e_inputs(int(g["W"]), int(g["N"]))
        v1 = workload != "c3"
        an = (pipeline.SLRv1Animator() if v1 else pipeline.BaselineAnimator())
        for name in (("encoder", "projector", "net_bg", "net_alpha_encoder", "net_alpha_decoder") if v1 else ("encoder", "projector")):
            keys = [str(k) for k in gn[f"{name}_keys"]]
            pre = NF.NETS[name][0]
            sd = {pre + k: v for k, v in NF.state_dict(name, keys, gn[f"{name}_shapes"]).items()}
            nets.load_reference_state_dict(getattr(an, name), sd, pre)
        an = an.to(dev).eval()
        ts = [int(t) for t in g["v1_ts"]] if v1 else list(range(n))
        fr = an.synthesize(torch.from_numpy(img).to(dev), torch.from_numpy(mo).to(dev), n, frames=ts).cpu().numpy()
        ref = g["v1_PredImg"] if v1 else g["baseline_PredImg"]
        out["frames_vs_reference_models_max_abs"] = float(np.abs(fr - ref).max())
    # (4) the same at 256 x 256 (the multi-tile / channel-blocked convolution variants of the timed clip), against digests of
    # the reference models' frames (tests/golden/large_nets_e2e.npz, tools/make_golden_large.py)
    lpath = os.path.join(ROOT, "tests", "golden", "large_nets_e2e.npz")
    if os.path.exists(epath) and os.path.exists(lpath):
        gl = np.load(lpath)
        S_, n2 = int(gl["S"]), int(gl["N"])
        img2, mo2, _ = NF.e2e_inputs(S_, n2)
        kind = "v1" if workload != "c3" else "baseline"
        ts2 = [n2 // 2] if kind == "v1" else [int(t) for t in gl["ts"]]
        fr2 = an.synthesize(torch.from_numpy(img2).to(dev), torch.from_numpy(mo2).to(dev), n2, frames=ts2).cpu().numpy()
        worst = 0.0
        for k2, t2 in enumerate(ts2):
            tag = f"{kind}_PredImg_t{t2}"
            pos = NF.digest_positions(tag, fr2[k2:k2 + 1].size, int(gl["npos"]))
            worst = max(worst, float(np.abs(fr2[k2:k2 + 1].ravel()[pos] - gl[f"{tag}_val"]).max()))
        out["frames_256_vs_reference_models_digest_max_abs"] = worst
    out["ok"] = bool(max(v for k, v in out.items() if k.endswith("max_abs")) < out["tolerance"])
    return out


# ------------------------------------------------------------------------------------------------ context (N = 1)

def _time_calls(fn, n, warm=3):
    for _ in range(warm):
        fn()
    evs = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    us = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in evs)
    return sum(us) / len(us), us[0]


@torch.no_grad()
def context_measurements(workload, image, motion, dev):
    import slr_sfs_amd as S
    from slr_sfs_amd import pipeline, synthesis
    out = {}
    # ---- splat stage alone (no networks): per clip Euler + row lists + plan, per frame the fused clip kernel
    fs = torch.randn(1, 64, H, W, device=dev)
    Z = torch.randn(1, 1, H, W, device=dev)
    best = 0.0
    for _ in range(3):                                       # the first clip allocates the plan buffers
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        cs = synthesis.ClipSynthesizer(fs, Z, motion, NFRAMES)
        B = synthesis.MAX_BATCH
        g = torch.empty(B, 64, H, W, device=dev)
        for t0 in range(0, NFRAMES, B):                      # as the pipelines do: up to 8 frames per launch
            ts = list(range(t0, min(t0 + B, NFRAMES)))
            cs.features_batch(ts, g[:len(ts)])
        torch.cuda.synchronize()
        best = max(best, NFRAMES / (time.perf_counter() - t1))
    out["splat_stage_fps_1gpu"] = round(best, 1)
    del g, cs
    out["roofline_dropin"] = dropin_roofline(dev, motion)
    out["roofline_conv"] = conv_roofline(dev)
    out["roofline_backward"] = backward_roofline(dev, motion)
    # ---- the other pipeline of BASELINE.json (C4 when C3 is timed and vice versa), one warm-up + two clips
    other = "c4" if workload == "c3" else "c3"
    torch.manual_seed(0)
    m2 = (pipeline.SLRv1Animator() if other == "c4" else pipeline.BaselineAnimator()).to(dev).eval()
    step = make_step(m2, image, motion, 0, 1, "rounds", "banded")
    osteps = 5
    dt, _, kev, sev = timed_clips(step, osteps, 1, 1, dev)
    c2 = 67 if other == "c4" else 65
    out[other] = {"value": round(NFRAMES * osteps / dt, 3), "unit": "frames/s", "steps": osteps, "warmup": 1,
                  "parity_err": parity_check(m2, image, motion, other, dev),
                  "workload": ("C4 SLR-v1 2-layer pipeline (fluid + background + alpha), " if other == "c4" else
                               "C3 baseline pipeline, ") + "768x1280, N=60",
                  "roofline": splat_roofline(kev, sev, c2, "slr::clip_tile_kernel<G2,false,true> (" +
                                             ("64 features + the alpha group" if other == "c4" else "64 features") + ")")}
    del m2
    # ---- the all-fp32 context: the same C3 clip with every convolution in the reference's arithmetic (fp32 operands, products and
    # accumulation) -- on this package's fp32 matrix-core kernels (convs="fp32"), and through PyTorch-ROCm (MIOpen fp32, convs="torch")
    if workload == "c3":
        out["roofline_conv_fp32"] = conv_roofline(dev, fp32=True)
        torch.manual_seed(0)
        m3 = pipeline.BaselineAnimator(convs="fp32").to(dev).eval()
        m3.synthesize(image, motion, NFRAMES)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        c32 = m3.synthesize(image, motion, NFRAMES)
        torch.cuda.synchronize()
        dt32 = time.perf_counter() - t1
        m3.convs = "fp32-winograd"                                                 # the same rung with Winograd 3x3 layers
        m3.synthesize(image, motion, NFRAMES)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        cw = m3.synthesize(image, motion, NFRAMES)
        torch.cuda.synchronize()
        dt32w = time.perf_counter() - t1
        dw = (cw - c32).abs()
        from slr_sfs_amd import nets as _nets
        m3.convs = "split"                                                         # (inside torch_convolutions() no kernel of ours runs: nothing to clamp)
        with _nets.torch_convolutions():                                           # the validation route, timed beside the product's fp32 rung
            m3.synthesize(image, motion, NFRAMES, frames=range(0, 6), batch=1)    # MIOpen picks its kernels
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            ct = m3.synthesize(image, motion, NFRAMES, frames=range(0, NFRAMES, 3), batch=1)
            torch.cuda.synchronize()
            dtt = time.perf_counter() - t1
        out["fps_fp32_convs"] = {"value": round(NFRAMES / dt32, 2), "unit": "frames/s",
                                 "what": "the same C3 clip (all 60 frames, one warm-up clip) with BaselineAnimator(convs='fp32'): every 3x3 / "
                                         "1x1 convolution of the encoder / decoder on v_mfma_f32_32x32x2_f32 (fp32 operands, fp32 products, "
                                         "fp32 accumulation, direct implicit GEMM: the reference's arithmetic; csrc/conv.hip, SLR_CONV_F32), "
                                         "every other stage and the splat path unchanged",
                                 "winograd": {"value": round(NFRAMES / dt32w, 2), "unit": "frames/s",
                                              "what": "convs='fp32-winograd': the rung's 3x3 layers as Winograd F(2x2,3x3) on the same instructions "
                                                      "(csrc/conv_wino.hpp): 16/36 of the products, fp32 throughout, 2-4x the direct kernel's rounding "
                                                      "error per layer",
                                              "frames_vs_direct_rung": {"mean_abs": float(dw.mean()), "max_abs": float(dw.max()),
                                                                        "values_beyond_1e-4": int((dw > 1e-4).sum()), "values": dw.numel()}},
                                 "through_torch_miopen": {"value": round(20 / dtt, 2), "unit": "frames/s",
                                                          "what": "inside nets.torch_convolutions() (validation route; 20 of the 60 frames): F.conv2d -> MIOpen fp32, elementwise "
                                                                  "stages as torch ops"},
                                 "frames_fp32_kernels_vs_torch_max_abs": float((c32[::3] - ct).abs().max())}
        del cw, dw
        del m3
    return out


def _graph_call_us(fn, reps=20, iters=10):
    """GPU time of everything one call launches, without the host's launch pace: `reps` calls captured into ONE HIP
    graph, the replay timed with an event pair, divided by reps (median of `iters` replays)."""
    fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn()                                              # workspaces of this stream exist before the capture
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(reps):
                fn()
        g.replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    torch.cuda.synchronize()
    del g                                                 # the graph is gone: the workspaces pinned for it may go as well
    import slr_sfs_amd as S
    S._lib.clear_workspaces(include_captured=True)
    ts.sort()
    return ts[len(ts) // 2]


@torch.no_grad()
def dropin_roofline(dev, motion):
    """The operator the reference scripts reach unchanged -- ModuleSoftsplat('summation') / _FunctionSoftsplat
    (softsplat.py:157-202, 390-424): one flow, C = 65 planes, 768x1280, on Euler-integrated flows; algorithmic bytes
    B_sum = (2C+2)*H*W*4 = 519.0 MB per call.  `tile` = the tile kernel alone (events recorded by the library around that
    launch), `call` = GPU time of EVERYTHING the call launches (front end + tile kernel + the pass-by-pass launch), measured by replaying a
    HIP graph of 20 calls; `call_eager_us` = the same call issued from Python with an event pair per call (host launch pace
    included).  Plus config C2 of BASELINE.json as stated (64 channels, 256x480, softmax mode, incoherent and smooth
    flow) and two more small grids.  `front_end` names what a call takes by default (include/slr_splat.h:
    slr_splat_set_front_end: scan up to 1024 tiles, rows above); the other front end is timed beside it.  `traffic` (tile kernel, Euler
    flows and C2): static_traffic()."""
    import slr_sfs_amd as S
    from slr_sfs_amd import synthesis
    L = S._lib.lib()
    C = 65
    x = torch.randn(1, C, H, W, device=dev)
    alg = (2 * C + 2) * H * W * 4
    res = {"bound": "hbm", "kernel": "slr::op_rows_kernel<false,false,false> (rows front end, the default at 1920 tiles); scan_front_end: "
                                     "slr::op_scan_kernel<false,false>", "peak": HBM_PEAK_GBS,
           "unit": "GB/s", "alg_bytes_per_call": alg, "flows": {},
           "call": "call_us = call_graph_us = GPU time of all launches of one call (HIP graph of 20 calls replayed; since round 3 -- rounds 1 and 2 "
                   "printed the eager figure under call_us); call_eager_us: Python call, event pair per call"}

    def measure(f, alg_bytes, tile=True):
        r = {}
        if tile:
            synthesis.kernel_timing = []
            for _ in range(12):
                synthesis._arm_timer(x)
                f()
            torch.cuda.synchronize()
            kus = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1, _ in synthesis.kernel_timing[3:])
            synthesis.kernel_timing = None
            k_avg = sum(kus) / len(kus)
            r.update({"tile_us": round(k_avg, 1), "tile_gbs": round(alg_bytes / k_avg / 1e3, 1),
                      "tile_frac": round(alg_bytes / k_avg / 1e3 / HBM_PEAK_GBS, 4)})
        eager, _ = _time_calls(f, 20)
        try:
            call = _graph_call_us(f)
        except Exception as e:                           # (a capture that fails must not cost the line: eager timing instead)
            call, r["call_note"] = eager, f"graph capture failed ({type(e).__name__}): call_us is the eager figure"
            torch.cuda.synchronize()
        r.update({"call_us": round(call, 1), "call_graph_us": round(call, 1), "call_frac": round(alg_bytes / call / 1e3 / HBM_PEAK_GBS, 4),
                  "call_eager_us": round(eager, 1), "call_eager_frac": round(alg_bytes / eager / 1e3 / HBM_PEAK_GBS, 4)})
        return r

    worst = None
    res["train_shape"] = train_shape_roofline(dev)
    flows = [("euler_t30", S.euler_integration(motion, 30)[0]), ("euler_t59", S.euler_integration(motion, 59)[0]),
             ("identity", torch.zeros(1, 2, H, W, device=dev)), ("incoherent", torch.rand(1, 2, H, W, device=dev) * 16 - 8)]
    for name, flow in flows:
        r = measure(lambda: S.FunctionSoftsplat(x, flow, None, "summation"), alg)
        r["front_end"] = "rows"                            # 1920 tiles > the scan threshold (1024): rowbin (+ plan) -> tile kernel -> deferred pieces
        if name.startswith("euler"):
            r["traffic"], r["traffic_source"] = static_traffic("op_rows_" + name[6:], 1)
            if r["traffic"]:                               # the tile kernel's physical share of the 8 TB/s (counter bytes / its time)
                r["frac_traffic"] = round(r["traffic"] / r["tile_us"] / 1e3 / HBM_PEAK_GBS, 4)
        prev = L.slr_splat_set_front_end(1)
        try:
            r["scan_front_end"] = measure(lambda: S.FunctionSoftsplat(x, flow, None, "summation"), alg)
        finally:
            L.slr_splat_set_front_end(prev)
        res["flows"][name] = r
        if name.startswith("euler") and (worst is None or r["call_frac"] < worst["call_frac"]):
            worst = r
    # the object's own figure is the CALL a user makes (all launches of the slower Euler flow), not the tile kernel alone
    res["achieved"], res["frac"], res["avg_us"] = round(alg / worst["call_us"] / 1e3, 1), worst["call_frac"], worst["call_us"]
    res["frac_is"] = "call_frac of the slower Euler flow (whole call, all launches); tile_frac / tile_us beside it per flow"
    res["tile_frac"], res["tile_us"] = worst["tile_frac"], worst["tile_us"]
    # config C2 of BASELINE.json: random 64-channel 256x480 features + flow, softmax mode, one call
    small = {}
    for tag, (c2, h2, w2) in (("c2", (64, 256, 480)), ("128x240", (64, 128, 240)), ("384x640", (65, 384, 640))):
        f2 = torch.randn(1, c2, h2, w2, device=dev)
        met = torch.randn(1, 1, h2, w2, device=dev)
        alg2 = (2 * c2 + 3) * h2 * w2 * 4
        cases = {"incoherent": torch.rand(1, 2, h2, w2, device=dev) * 16 - 8}
        if tag in ("c2", "128x240"):                 # Euler-integrated smooth fields: the flows the reference produces (pile-ups -> the sink launch)
            cases["smooth_t30"] = S.euler_integration(torch.from_numpy(smooth_motion(h2, w2)).to(dev), 30)[0]
        if tag == "c2":
            cases["smooth_t59"] = S.euler_integration(torch.from_numpy(smooth_motion(h2, w2)).to(dev), 59)[0]
        for fname, fl2 in cases.items():
            r = measure(lambda: S.FunctionSoftsplat(f2, fl2, met, "softmax"), alg2, tile=False)
            r.update({"workload": f"FunctionSoftsplat softmax, {c2} ch, {h2}x{w2}, {fname} flow", "alg_bytes": alg2,
                      "front_end": "scan (box kernel + tile kernel + the sink launch)"})
            prev = L.slr_splat_set_front_end(2)
            try:
                try:
                    r["rows_front_end_call_us"] = round(_graph_call_us(lambda: S.FunctionSoftsplat(f2, fl2, met, "softmax")), 1)
                except Exception:
                    r["rows_front_end_call_us"] = round(_time_calls(lambda: S.FunctionSoftsplat(f2, fl2, met, "softmax"), 20)[0], 1)
            finally:
                L.slr_splat_set_front_end(prev)
            small[tag if fname == "incoherent" else f"{tag}_{fname}"] = r
    res["c2"] = small.pop("c2")
    res["c2"]["workload"] = "C2: " + res["c2"]["workload"] + " U(-8,8)"
    res["c2"]["traffic"], res["c2"]["traffic_source"] = static_traffic("op_scan_c2", 1)
    if res["c2"]["traffic"]:
        res["c2"]["frac_traffic"] = round(res["c2"]["traffic"] / res["c2"]["call_us"] / 1e3 / HBM_PEAK_GBS, 4)
    res["small_grids"] = small
    # config C2 with N = 60 as ONE call (BASELINE.json: "random 64-ch 256x480 feature + flow, N=60"): FunctionSoftsplat on
    # [60,64,256,480] with 60 different incoherent flows -- 14400 output tiles, the rows front end, the chip full
    nb = 60
    fb = torch.randn(nb, 64, 256, 480, device=dev)
    mb = torch.randn(nb, 1, 256, 480, device=dev)
    flb = torch.rand(nb, 2, 256, 480, device=dev) * 16 - 8
    algb = nb * (2 * 64 + 3) * 256 * 480 * 4
    rb = measure(lambda: S.FunctionSoftsplat(fb, flb, mb, "softmax"), algb, tile=False)
    rb.update({"workload": "C2 batched: ONE FunctionSoftsplat(softmax) call on [60,64,256,480], 60 incoherent U(-8,8) flows", "alg_bytes": algb,
               "front_end": "rows", "per_sample_us": round(rb["call_us"] / nb, 2)})
    res["c2_batched"] = rb
    del fb, mb, flb
    return res


def train_shape_roofline(dev):
    """The shape the reference TRAINS the operator at: W = 256, batch 2 per GPU (train_animating_scripts/train_baseline2_pconv.sh:14,
    options/train_options.py:271) -- _FunctionSoftsplat forward + backward (both gradients) on [2,65,256,256] with Euler-integrated
    smooth flows, t = 30 and t = 59 (models/animating_softmax_splating.py:579-595: 64 features + the weight plane through
    ModuleSoftsplat('summation'), softsplat.py:390-479).  fwd_us / bwd_us: GPU time of everything the forward call / the backward call
    launches (20 calls in a HIP graph); bytes: forward B_sum = (2C+2) N H W 4, backward (3C+4) N H W 4."""
    import slr_sfs_amd as S
    from slr_sfs_amd._lib import check, lib, ptr, stream_of
    L = lib()
    N, C, h, w = 2, 65, 256, 256
    x, go = torch.randn(N, C, h, w, device=dev), torch.randn(N, C, h, w, device=dev)
    gi, gf = torch.empty_like(x), torch.empty(N, 2, h, w, device=dev)
    mo = torch.from_numpy(np.concatenate([smooth_motion(h, w, seed=0), smooth_motion(h, w, seed=1)], 0)).to(dev)
    alg_f, alg_b = (2 * C + 2) * N * h * w * 4, (3 * C + 4) * N * h * w * 4
    res = {"bound": "hbm", "shape": [N, C, h, w], "peak": HBM_PEAK_GBS, "unit": "GB/s", "alg_bytes_forward": alg_f, "alg_bytes_backward": alg_b,
           "what": "_FunctionSoftsplat forward + backward (gradInput + gradFlow) at the reference's training shape, Euler flows of two smooth fields",
           "flows": {}}
    for t in (30, 59):
        fl = S.EulerIntegration()(mo, torch.tensor([t, t], device=dev)).contiguous()
        with torch.no_grad():
            fwd = _graph_call_us(lambda: S.softsplat._FunctionSoftsplat.apply(x, fl))
            nb = int(L.slr_softsplat_backward_ws_bytes(N, C, h, w))          # (channel groups on small grids: what the autograd route uses)
            bws = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)
            bwd = _graph_call_us(lambda: check(L.slr_softsplat_backward_ws(ptr(x), ptr(fl), ptr(go), ptr(gi), ptr(gf), N, C, h, w, ptr(bws), nb, stream_of(x)), "backward"))
        # the autograd route end to end (eager, host launch pace included): what a training step pays
        xg, fg = x.clone().requires_grad_(True), fl.clone().requires_grad_(True)

        def step():
            with torch.enable_grad():
                out = S.softsplat._FunctionSoftsplat.apply(xg, fg)
                torch.autograd.grad(out, (xg, fg), go)
        eager, _ = _time_calls(step, 20)
        res["flows"][f"euler_t{t}"] = {"fwd_us": round(fwd, 1), "bwd_us": round(bwd, 1), "fwd_frac": round(alg_f / fwd / 1e3 / HBM_PEAK_GBS, 4),
                                       "bwd_frac": round(alg_b / bwd / 1e3 / HBM_PEAK_GBS, 4),
                                       "frac": round((alg_f + alg_b) / (fwd + bwd) / 1e3 / HBM_PEAK_GBS, 4), "autograd_eager_us": round(eager, 1)}
    return res


@torch.no_grad()
def backward_roofline(dev, motion):
    """The reference's backward kernels (softsplat.py:204-326) as ONE gather kernel, 65 planes at 768x1280 on Euler-integrated flows:
    gradInput + gradFlow in one launch; algorithmic bytes = gradOutput + input read, gradInput written = 3*C*H*W*4 (+ flow, gradFlow).
    avg_us: the launches of a call (slr_softsplat_backward_ws: the gather kernel + the 8 MB sum of the second group's partial gradFlow), 20 calls captured into a HIP graph and replayed (event pair around the replay); eager_us: event pair
    around each Python call."""
    import slr_sfs_amd as S
    from slr_sfs_amd._lib import check, lib, ptr, stream_of
    L = lib()
    C = 65
    x, go = torch.randn(1, C, H, W, device=dev), torch.randn(1, C, H, W, device=dev)
    gi, gf = torch.empty_like(x), torch.empty(1, 2, H, W, device=dev)
    nb = int(L.slr_softsplat_backward_ws_bytes(1, C, H, W))              # (two channel groups: what the autograd route uses)
    bws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
    alg = (3 * C + 4) * H * W * 4
    res = {"bound": "hbm", "kernel": "slr::grad_tile_kernel<true,true> (gradInput + gradFlow, two channel groups) + grad_flow_sum_kernel", "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "alg_bytes_per_launch": alg, "flows": {}}
    for name, t in (("euler_t30", 30), ("euler_t59", 59), ("identity", 0)):
        fl = S.euler_integration(motion, t)[0] if t else torch.zeros(1, 2, H, W, device=dev)
        # (the library launches on the stream it is handed: inside _graph_call_us that is the capturing stream)
        call = lambda: check(L.slr_softsplat_backward_ws(ptr(x), ptr(fl), ptr(go), ptr(gi), ptr(gf), 1, C, H, W, ptr(bws), nb, stream_of(x)), "backward")
        eager, _ = _time_calls(call, 20)
        avg = _graph_call_us(call)
        r = {"avg_us": round(avg, 1), "eager_us": round(eager, 1), "achieved": round(alg / avg / 1e3, 1), "frac": round(alg / avg / 1e3 / HBM_PEAK_GBS, 4)}
        if name == "euler_t30":
            r["traffic"], r["traffic_source"] = static_traffic("grad_t30", 1)
            res.update({"achieved": r["achieved"], "frac": r["frac"], "avg_us": r["avg_us"], "traffic": r["traffic"]})
        res["flows"][name] = r
    return res


def conv_roofline(dev, fp32=False):
    """Second kernel of the frame (and since the splat is fused, the dominant one by time): the
    matrix-core partial convolution, timed with HIP events on its launch stream (torch's current
    stream) on the decoder's heaviest layer shape, in the CHANNEL-BLOCKED instantiation the decoder actually
    runs (activations [N, C/8, H, W, 8] in and out).  `achieved` counts the ALGORITHMIC flops of the
    convolution (2*9*Cin*Cout*H*W); the kernel issues 3 f16 MFMAs per product (split operands),
    `issued` = 3 x achieved is what the matrix pipe executes; peak = dense f16 MFMA rate."""
    from slr_sfs_amd import nets
    cin = cout = 128
    pc = nets.PartialConv(cin, cout, 3).to(dev)
    x = torch.randn(1, cin, H, W, device=dev)            # any values: read as [1, C/8, H, W, 8]
    mask = (torch.rand(1, 1, H, W, device=dev) > 0.1).float()
    sc, sh = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.3
    nb = (torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.3)
    lay = nets.IN_B8 | nets.OUT_B8
    flops = 2.0 * 9 * cin * cout * H * W
    if fp32:                                             # the fp32 rung: same call, same layouts, products on v_mfma_f32_32x32x2_f32
        with torch.no_grad(), nets.fp32_kernels(winograd=False):
            avg_d, mn_d = _time_calls(lambda: pc(x, mask, next_bn=nb, pre_bn=(sc, sh), layout=lay), 15, warm=3)
        with torch.no_grad(), nets.fp32_kernels(winograd=True):       # convs='fp32-winograd': 16 instead of 36 products per 2x2 outputs
            avg, mn = _time_calls(lambda: pc(x, mask, next_bn=nb, pre_bn=(sc, sh), layout=lay), 15, warm=3)
        issued = flops * 16.0 / 36.0                     # what the matrix pipe executes in the Winograd kernel
        ach_w = issued / (avg * 1e-6) / 1e12
        ach_d = flops / (avg_d * 1e-6) / 1e12
        return {"bound": "mfma", "kernel": "slr::conv3x3_split_kernel<1,4,true,true,F32> (128->128, 768x1280, channel-blocked in/out, BN+mask "
                                           "prologue, partial-conv epilogue + next BN) on v_mfma_f32_32x32x2_f32: the strict rung (convs='fp32')",
                "achieved": round(ach_d, 1), "peak": 157.3, "unit": "TFLOP/s", "frac": round(ach_d / 157.3, 4),
                "avg_us": round(avg_d, 1), "min_us": round(mn_d, 1), "launches": 15,
                "precision": "fp32 operands, fp32 products, fp32 accumulation (one MFMA per product): the reference's arithmetic",
                "winograd": {"kernel": "slr::conv3x3_wino_kernel<true,true> (same layer; Winograd F(2x2,3x3) on the same instructions, "
                                       "csrc/conv_wino.hpp: convs='fp32-winograd')",
                             "achieved": round(ach_w, 1), "frac": round(ach_w / 157.3, 4),
                             "achieved_is": "the flops the kernel ISSUES to the matrix pipe (16/36 of 2*9*Cin*Cout*H*W) per second",
                             "direct_equivalent": round(flops / (avg * 1e-6) / 1e12, 1), "avg_us": round(avg, 1), "min_us": round(mn, 1),
                             "precision": "fp32 operands, products, accumulation; transforms are fp32 additions; error per layer vs fp64 <= 1.1e-6 "
                                          "of the output range (direct: 3e-7), tests/test_gpu_conv_f32.py"}}
    with torch.no_grad():
        avg, mn = _time_calls(lambda: pc(x, mask, next_bn=nb, pre_bn=(sc, sh), layout=lay), 30, warm=5)
    ach = flops / (avg * 1e-6) / 1e12
    return {"bound": "mfma", "kernel": "slr::conv3x3_split_kernel<1,4,true,true> (128->128, 768x1280, channel-blocked in/out, "
                                       "BN+mask prologue, partial-conv epilogue + next BN: the variant 14 of the decoder's 16 "
                                       "convolutions run)",
            "achieved": round(ach, 1), "issued": round(3 * ach, 1), "peak": 2500.0, "unit": "TFLOP/s",
            "frac": round(ach / 2500.0, 4), "frac_issued": round(3 * ach / 2500.0, 4),
            "fp32_mfma_peak": 157.3, "avg_us": round(avg, 1), "min_us": round(mn, 1), "launches": 30,
            "precision": "fp32 in/out, fp32 accumulation; operands split into two f16 halves (22 significant bits), "
                         "3 MFMAs per product; measured whole-decoder max-abs error vs fp64 1.0e-5 "
                         "(torch fp32: 0.5e-5); vs the reference's own decoder class <= 5e-5 of the output range "
                         "(tests/test_nets_golden.py)"}


def cpu_baseline(motion):
    """The CPU oracle on the host cores: same hot path (Euler + 2 x 65-plane splat + normalise per frame), a bounded
    sample of frames with all cores, and a smaller one with ONE thread (BASELINE.md section 4 asks for both)."""
    from oracle import oracle as o          # test infrastructure, used here only as the timed baseline
    o.build()
    rng = np.random.default_rng(1)
    fs = rng.standard_normal((1, 64, H, W)).astype(np.float32)
    Z = rng.standard_normal((1, 1, H, W)).astype(np.float32)
    cores = o.max_threads()
    frames = tuple(range(0, NFRAMES, 2))            # 30 frames: a bounded sample of the clip
    t0 = time.perf_counter()
    for t in frames:
        o.synth_baseline(fs, Z, motion, t, NFRAMES)
    dt = time.perf_counter() - t0
    one = (10, 30, 50)
    o.set_threads(1)
    t0 = time.perf_counter()
    for t in one:
        o.synth_baseline(fs, Z, motion, t, NFRAMES)
    dt1 = time.perf_counter() - t0
    o.set_threads(cores)
    # the networks either side of the path on the same cores: the reference's decoder formulation through torch's CPU
    # convolutions (the package's torch definition of the nets, which the tests hold against the reference's classes)
    from slr_sfs_amd import nets
    gen = torch.from_numpy(o.synth_baseline(fs, Z, motion, 30, NFRAMES))
    with nets.cpu_reference(), torch.no_grad():
        dec = nets.DecoderPconv2(64, 3).eval()
        t0 = time.perf_counter()
        torch.tanh(dec(gen))
        ddec = time.perf_counter() - t0
    whole = 1.0 / (dt / len(frames) + ddec)
    return {"value": round(len(frames) / dt, 4), "unit": "frames/s (splat stage: Euler + 2x65-plane splat + normalise; "
            "no encoder/decoder)", "cores": cores, "kind": "port",
            "sample": f"{len(frames)} frames (every 2nd) of the same 768x1280 N=60 clip, {dt:.1f} s of CPU work",
            "one_thread": {"value": round(len(one) / dt1, 4), "cores": 1,
                           "sample": f"frames {one} of the same clip, {dt1:.1f} s of CPU work"},
            "with_decoder": {"value": round(whole, 4), "unit": "frames/s (splat stage + partial-conv decoder per frame; encoder "
                             "once per clip not counted)", "decoder_s_per_frame": round(ddec, 3),
                             "torch_threads": torch.get_num_threads(),
                             "sample": "one frame through the decoder (torch CPU convolutions, fp32)"}}


if __name__ == "__main__":
    main()
