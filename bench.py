#!/usr/bin/env python3
"""bench.py -- Mrays/s (primary + shadow rays actually cast) of the render path on N MI355X.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over the workload: `spp` progressive samples of the
whole frame at max_depth = 0 (rays = primary + shadow), with the
scene resident in HBM.  With N > 1 the frame is tile-partitioned over the ranks (64x64
tiles, round-robin; scene replicated) and the step ends with the RCCL gather of the
partition framebuffers -- total work is fixed, so scaling is "strong".

The default workload is C3 (BASELINE.json configs[2]): the largest single-GPU 1080p
configuration and the one the metric's "primary+shadow" wording describes (69 451
triangles, 3 lights with shadow rays, 64 spp).  `--config c2` runs configs[1] (Cornell
box, 36 triangles, 16 spp), whose whole step is a few milliseconds.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel: over a deep tree
k_batch (camera rays, shadow rays and running mean of a whole batch of samples, one launch
per batch and stream), over a shallow one k_path (the whole path of a pixel-sample in one
launch).  Algorithmic bytes per launch / its mean HIP-event duration (events on the stream
the kernel runs on), against the 8 TB/s HBM peak.  Unit = one pixel-sample.  Algorithmic
bytes per unit ("Measurement"): k_batch moves nothing per sample -- per pixel and
launch the framebuffer texel (read + write), the RNG counter and draw count (read + write),
the pixel-map entry and the id / t buffers, 60 B; k_path 52 B per pixel-sample = framebuffer
texel read + write (32), RNG counter read + write (8), pixel-map entry (4), id + t buffers
(8); plus one pass over the scene records per launch (64 B triangle record + 32 B of box
tree per triangle, 132 B per sphere).  The figure SURVEY 8(d) defined for a plain scan (every
ray streams every triangle, one fetch shared by 256 rays) is reported next to it as
`scan_equivalent`: it is what a scan would have had to move to produce the same result in
the same time.  `cpu_baseline` is the CPU oracle (a port of the reference's per-pixel code)
timed on this box's host cores on a bounded crop of the same workload.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
TRI_BYTES = 36.0             # 9 floats of positions per triangle (scan model)
TRI_REC_BYTES = 64.0 + 32.0  # culled kernel: TriRec + its share of the box tree (24 B * 4/3)
SPH_BYTES = 132.0            # mat4 + inverse + radius
RAYS_PER_FETCH = 256.0       # one workgroup of 256 rays shares a fetch
RAY_IO_BYTES = 32.0          # scan model only: 24 B ray in + 8 B (t, id) out per ray
PIXEL_SAMPLE_BYTES = 52.0    # k_path: fb 16+16, rng counter 4+4, pixmap 4, first id + t 8
VALU_MIX = "profiles/r03_valu_mix.json"   # per launch: the kernel's dynamic VALU instruction classes priced with the SIMD
                             # cycles tools/valu_microbench.hip measured (2 simple / 4 complex or packed / 8 transcendental;
                             # tools/valu_ceiling.py).  lo / hi: what the counters cannot split priced both ways.
BATCH_PIXEL_BYTES = 60.0     # k_batch, per pixel and launch: fb 16+16, rng counter 4+4, last draws 4+4, pixmap 4, id + t 8
PROFILE_ROUNDS = ("r03", "r02", "r01")   # committed rocprofv3 summaries under profiles/, newest first


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c3", choices=["c1", "c2", "c3", "c4", "c5"])
    ap.add_argument("--spp", type=int, default=0, help="samples per step (default: the config's)")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearsal of the N > 1 path with every rank on ONE GPU "
                         "(gather staged through host memory); not a measurement")
    ap.add_argument("--collective-with-one-rank", action="store_true",
                    help="with one GPU: run the N > 1 branch (process group, export, all-gather, "
                         "all-reduces) with a world of one -- a rehearsal of that branch, not a bench line")
    ap.add_argument("--cpu-crop", type=int, nargs=2, default=None, metavar=("W", "H"))
    return ap.parse_args()


def measured_traffic(cfg_name, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes
    (profiles/r01_traffic.json, written by tools/pmc_traffic.py on the GPU box: FETCH_SIZE and
    WRITE_SIZE in separate passes, FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 note).
    None if there is no profile for this workload."""
    for rnd in PROFILE_ROUNDS:
        path = os.path.join(ROOT, "profiles", f"{rnd}_traffic.json")
        try:
            with open(path) as f:
                d = json.load(f)
            return float(d[cfg_name][kernel]["bytes_per_launch"])
        except (OSError, KeyError, ValueError, TypeError):
            continue
    return None


def measured_counter(cfg_name, kernel, counter):
    """Per-launch value of an SQ counter of `kernel` from the committed PMC summary
    (profiles/r01_<cfg>_pmc_sq.txt, written by tools/pmc_summary.py: one "== kernel" section
    per kernel).  None if absent."""
    for rnd in PROFILE_ROUNDS:
        path = os.path.join(ROOT, "profiles", f"{rnd}_{cfg_name}_pmc_sq.txt")
        try:
            section = None
            for line in open(path):
                f = line.split()
                if f and f[0] == "==":
                    section = f[1]
                elif f and section == kernel and f[0] == counter:
                    return float(f[f.index("launch") + 1])
        except (OSError, ValueError, IndexError):
            pass
    return None


def valu_mix(cfg_name, kernel):
    """Mix-weighted VALU pipe cycles per launch of `kernel` (tools/valu_ceiling.py), None if absent."""
    try:
        with open(os.path.join(ROOT, VALU_MIX)) as f:
            return json.load(f)[cfg_name][kernel]
    except (OSError, KeyError, ValueError, TypeError):
        return None


def cpu_baseline(cfg, spp, crop_wh):
    """Oracle on host cores, bounded sample (rank 0, N = 1 only)."""
    from tests import orc
    cw, ch = crop_wh
    cw, ch = min(cw, cfg.width), min(ch, cfg.height)
    x0, y0 = (cfg.width - cw) // 2, (cfg.height - ch) // 2
    threads = min(orc.lib().orc_num_threads(), 16)
    t0 = time.perf_counter()
    r = orc.render(cfg.scene, cfg.camera, cfg.width, cfg.height, spp=spp, max_depth=0,
                   crop=(x0, y0, x0 + cw, y0 + ch), threads=threads)
    dt = time.perf_counter() - t0
    rays = r["primary"] + r["shadow"]
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"oracle, centre crop {cw}x{ch} of {cfg.width}x{cfg.height}, {spp} spp, "
                      f"max_depth 0, {rays} rays in {dt:.2f} s"}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")

    import numpy as np
    import torch
    from raytracer_amd import abi, scenes

    # --collective-with-one-rank: a rehearsal of the N > 1 branch where only one GPU is at hand -- the
    # process group, the export into the collective's buffer, the all-gather, the barrier and the
    # all-reduces all run, over RCCL, with a world of one
    multi = world > 1 or a.collective_with_one_rank
    dist = None
    if multi:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if a.backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    dev = local_rank if (multi and a.backend == "nccl") else 0
    cdev = "cuda" if a.backend == "nccl" else "cpu"      # where the collectives' tensors live
    torch.cuda.set_device(dev)

    kw = {}
    if a.width and a.height:
        kw = dict(width=a.width, height=a.height)
    cfg = scenes.CONFIGS[a.config](**kw)
    spp = a.spp or cfg.spp

    ctx = abi.Context(cfg.width, cfg.height, dev)
    ctx.upload_scene(cfg.scene)
    ctx.set_camera(cfg.camera)
    ctx.set_partition(rank, world, 64)
    n_local = ctx.local_pixel_count()

    gather = None
    if multi:
        from raytracer_amd.dist import FrameGather
        gather = FrameGather(cfg.width, cfg.height, world, 64, cdev, abi.partition_pixels)
        stage = torch.zeros_like(gather.local, device="cuda") if cdev == "cpu" else None

    def step():
        ctx.reset()                      # a fresh frame: counters, RNG streams, framebuffer
        ctx.render(spp, 0)
        if multi:                        # final framebuffer gathered over xGMI (RCCL)
            # the previous step's gather (asynchronous, on torch's stream) still reads gather.local: it has had
            # a whole render to finish, this makes it certain before the buffer is overwritten
            torch.cuda.current_stream().synchronize()
            if stage is None:
                ctx.export_local_rgba(gather.local.data_ptr())
            else:
                ctx.export_local_rgba(stage.data_ptr())
                gather.local.copy_(stage)
            gather.gather(dist)

    def fence():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    ctx.reset_stats()
    ctx.set_timing(True)
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    st = ctx.stats()
    ctx.set_timing(False)

    rays_local = float(st.primary_rays + st.shadow_rays)
    if multi:
        tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        rr = torch.tensor([rays_local], dtype=torch.float64, device=cdev)
        dist.all_reduce(rr, op=dist.ReduceOp.SUM)
        rays_total = float(rr.item())
    else:
        rays_total = rays_local

    if rank == 0:
        sc = cfg.scene
        n_tri, n_sph = sc.n_tris, sc.n_quads
        # dominant kernel of the step: k_batch (depth 0 over a deep tree: a whole batch of samples)
        # or k_path (everything of one sample).  Unit of a launch = pixel-samples.
        fused = st.batch_launches > 0
        kernel = "k_batch" if fused else "k_path"
        rays_k = float(st.primary_rays + st.shadow_rays + st.bounce_rays)   # this rank
        launches = max(1, st.path_launches)
        k_ms = st.path_ms / launches
        units_per_launch = float(st.samples) / launches
        n_pixels_rank = float(st.samples) / max(1, spp * a.steps)   # stats cover the timed steps
        if fused:      # per pixel and launch only; nothing per sample reaches memory
            unit_bytes = BATCH_PIXEL_BYTES * (n_pixels_rank / max(1, int(st.streams))) / max(units_per_launch, 1.0)
        else:
            unit_bytes = PIXEL_SAMPLE_BYTES
        rays_per_launch = rays_k / launches
        bytes_per_launch = (unit_bytes * units_per_launch + n_tri * TRI_REC_BYTES
                            + n_sph * SPH_BYTES)
        achieved = bytes_per_launch / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        # what a scan would have had to move for the same result in the same time
        scan_per_ray = (n_tri * TRI_BYTES + n_sph * SPH_BYTES) / RAYS_PER_FETCH + RAY_IO_BYTES
        scan_gbs = scan_per_ray * (rays_total / a.steps) / (dt / a.steps) / 1e9 if dt > 0 else 0.0
        tests = rays_k * (n_tri + n_sph)
        out = {
            "metric": f"Mrays/sec (primary+shadow) at {cfg.height}p",
            "value": rays_total / dt / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{cfg.name}: {cfg.description}", "width": cfg.width,
                       "height": cfg.height, "spp_per_step": spp, "max_depth": 0,
                       "triangles": n_tri, "spheres": n_sph, "lights": sc.n_lights,
                       "partition": f"64x64 tiles round-robin over {world} GPU(s)",
                       "rays_per_step": rays_total / a.steps},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": measured_traffic(cfg.name, kernel),
                         "kernel": kernel, "avg_launch_ms": k_ms,
                         "launches_timed": int(launches),
                         "bytes_per_launch": bytes_per_launch,
                         "pixel_samples_per_launch": units_per_launch,
                         "rays_per_launch": rays_per_launch,
                         "concurrent_streams": int(st.streams),
                         "samples_per_launch_per_pixel": units_per_launch / max(1.0, n_pixels_rank / max(1, int(st.streams))) if fused else None,
                         "note": "the path is bound by VALU issue and scalar-load latency of the "
                                 "tree traversal, not by HBM (DESIGN.md); HBM fraction reported "
                                 "as asked"},
            "scan_equivalent": {"model": "SURVEY 8(d): every ray streams every primitive, one "
                                         "fetch per 256 rays",
                                "gbs": scan_gbs, "frac_of_hbm_peak": scan_gbs / HBM_PEAK_GBS,
                                "pair_tests_per_s": tests / dt if dt else 0.0},
        }
        # What the kernel is closest to: the VALU pipe.  Per launch, the cycles its instruction mix occupies a SIMD's
        # VALU for (committed counters x measured per-class cycles, tools/valu_ceiling.py) against the cycles the
        # launch had on every SIMD; the duration is live, and the streams' launches share the chip.
        mix = valu_mix(cfg.name, kernel) if spp == cfg.spp else None     # the counters were taken at the config's own spp
        if mix and k_ms > 0:
            prop = torch.cuda.get_device_properties(dev)
            simds = prop.multi_processor_count * 4
            clk_hz = float(getattr(prop, "clock_rate", 2400000)) * 1e3      # nominal: the clock under load is lower
            if fused:   # every stream's launches share the chip: all cycles of the step against its wall time
                avail = dt / a.steps * simds * clk_hz / (float(launches) / a.steps)
            else:
                avail = k_ms * 1e-3 * simds * clk_hz / max(1, int(st.streams))
            out["valu_issue"] = {"source": VALU_MIX + " (unpacked build's per-class counters, same arithmetic and speed)",
                                 "valu_insts_per_launch": measured_counter(cfg.name, kernel, "SQ_INSTS_VALU"),
                                 "pipe_cycles_per_launch_lo": mix["pipe_cycles_lo"], "pipe_cycles_per_launch_hi": mix["pipe_cycles_hi"],
                                 "simds": simds, "clock_hz": clk_hz,
                                 "frac_of_valu_pipe_lo": mix["pipe_cycles_lo"] / avail,
                                 "frac_of_valu_pipe_hi": mix["pipe_cycles_hi"] / avail,
                                 "note": "mix-weighted with measured prices (2 / 4 / 8 cycles per wave64 instruction), not a flat "
                                         "4 cycles per instruction; what is left is issue latency of four waves per SIMD (DESIGN.md 5)"}
        if a.collective_with_one_rank and world == 1:
            # the frame the all-gather assembled against the context's own framebuffer, bit for bit
            whole = gather.frame[:-1].view(cfg.height, cfg.width, 4).cpu().numpy()
            mine = ctx.read_framebuffer()["rgba"]
            out["collective_rehearsal"] = {
                "backend": a.backend, "world": 1,
                "frame_equals_framebuffer": bool(np.array_equal(whole.view(np.uint32), mine.view(np.uint32))),
                "note": "the N > 1 branch run with a world of one; not a scaling figure"}
        if world == 1 and not a.no_cpu:
            crop = tuple(a.cpu_crop) if a.cpu_crop else {"c1": (640, 480), "c2": (1600, 900),
                                                         "c3": (192, 108), "c4": (32, 18),
                                                         "c5": (8, 4)}[cfg.name]
            cpu_spp = spp if cfg.name in ("c1", "c2") else 1
            out["cpu_baseline"] = cpu_baseline(cfg, cpu_spp, crop)
        print(json.dumps(out), flush=True)

    ctx.close()
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
