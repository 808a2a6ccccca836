#!/usr/bin/env python3
"""Headline benchmark of the MI355X many-world ECS backend.

Metric (BASELINE.json): aggregate env steps/sec at N worlds + achieved HBM GB/s
on the sort node.  A "step" is one replay of the simulator's task graph over
all of the rank's worlds (one hipGraph launch on the executor's stream).

N=1 workload = BASELINE.json configs[2], the configuration the north_star target
is quoted on: Escape-Room + XPBD rigid bodies + BVH broadphase, 8192 worlds on
one MI355X (sims/escape_room_phys; synthetic worlds, random actions resident in
HBM, every world resets itself with probability 1/200 per step so that the
compaction sorts and BVH rebuilds run on live data every step).  configs[1]
(physics off, 4096 worlds) rides along as the secondary key `ecs_config2`.
N>1 runs the SAME workload (weak scaling: 8192 worlds per GPU, worlds sharded
by global index) with one packed all-gather of the observation tensors per step
over RCCL/xGMI, so that the points of a 1 -> 8 GPU curve compare.  BASELINE.json
configs[3] (Hide-and-Seek-shaped worlds, 8 x 8192) is `--sim hideseek`; its
one-GPU share rides along at N=1 as `hideseek_config4_share`, the Cartpole
plumbing case (configs[0]) as `cartpole_config1`.

    python bench.py --gpus 1 --steps 600 --warmup 100
    python bench.py --gpus 8 ...        # launches its own ranks, or:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \
        --master-addr 127.0.0.1 --master-port 29500 bench.py --gpus 8 ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
MIN_WINDOW_S = 0.5      # a timed window shorter than this is repeated over more steps
# instruction issue (the yardstick of the kernels that HBM says nothing about):
# 256 CUs x 4 SIMDs; a wavefront issues at most one instruction every ~4 cycles
# (MI355X_MICROARCH.md, "8 issue slots of ~4 cyc" per 32-cycle MFMA at one wave
# per SIMD), so ONE wave per SIMD -- what the LDS-resident physics step runs at
# -- tops out at 0.25 wave-instructions per cycle and SIMD; two waves per SIMD
# can interleave up to 0.5 (a wave64 VALU op occupies the SIMD-32 for 2 cycles)
NUM_SIMDS = 1024
SHADER_CLOCK_HZ = 2.4e9
ISSUE_PEAK_ONE_WAVE = 0.25
ISSUE_PEAK_SIMD = 0.5

OBS_TENSORS = {
    "escape_room": ["self_obs", "partner_obs", "room_ent_obs", "door_obs", "lidar",
                    "reward", "done", "steps_remaining"],
    "escape_room_phys": ["self_obs", "partner_obs", "room_ent_obs", "door_obs", "lidar",
                         "reward", "done", "steps_remaining"],
    "hideseek": ["self_obs", "agent_obs", "box_obs", "ramp_obs", "lidar",
                 "reward", "done"],
}

# rigid bodies per world: 2 agents + 23 PhysicsEntity + 3 doors; 5 agents + 11
# movable + 13 static
PHYS_BODIES = {"escape_room_phys": 28, "hideseek": 29}
# per body the fused step reads 156 B (transform, velocity, forces, ids, leaf +
# slot box) and writes 132 B (transform, velocity, solver state) -- DESIGN.md §10
PHYS_BYTES_PER_BODY = 288.0
# the leaf update + refit per body: 4 (WorldID) + LeafID 4, Position 12,
# Rotation 16, Scale 12, ObjectID 4, Velocity 24, object box 24 in; leaf box 24,
# leaf transform 40, slot box 24 out (systemIO<updateLeafAndRefitEntry>)
LEAF_REFRESH_BYTES_PER_BODY = 188.0
AGENTS = {"escape_room": 2, "escape_room_phys": 2, "hideseek": 5}

WORKLOADS = {
    "escape_room": (4096, "Escape-Room-shaped ECS (physics off), {w} worlds per GPU "
                          "(BASELINE.json configs[1]), 29 entity rows/world"),
    "escape_room_phys": (8192, "Escape-Room + XPBD rigid body + LBVH broadphase, {w} "
                               "worlds per GPU (BASELINE.json configs[2]), 28 rigid "
                               "bodies + 6 buttons/world, 4 substeps, grab joints"),
    "escape_room_render": (8192, "Escape-Room + XPBD + batch ray caster, {w} worlds "
                                 "(BASELINE.json configs[4])"),
    "hideseek": (8192, "Hide-and-Seek-shaped: XPBD + LBVH, {w} worlds per GPU "
                       "(BASELINE.json configs[3] = 8 x 8192), 29 rigid bodies/world "
                       "(5 agents, 9 boxes, 2 wedge ramps, 12 walls, plane), lock "
                       "action, line-of-sight rays, 30-ray lidar, 2.9 KB obs/world"),
}


LINE_BUDGET_BYTES = 8192    # the driver's parser lost round 4's 21 KB line


def _five(node):
    """Five-number summary of a roofline node for the driver line."""
    if not node:
        return None
    out = {"achieved": node.get("achieved"), "frac": node.get("frac"),
           "avg_us": node.get("avg_us"), "traffic": node.get("traffic"),
           "algo_bytes": node.get("algo_bytes_per_launch")}
    if "unit" in node and node["unit"] != "GB/s":
        out["unit"] = node["unit"]
        out["peak"] = node.get("peak")
    if "launches" in node:
        out["launches"] = node["launches"]
    return out


def _short(text, n=96):
    text = str(text)
    return text if len(text) <= n else text[:n - 1] + "~"


def driver_line(full, detail_path):
    """The ONE line the driver parses: the contract's keys, `roofline` (dominant
    kernel + five-number node summaries), `cpu_baseline`, one-number summaries of
    the other configs.  Everything else (per-kernel tables, chains, notes) is in
    `full`, which the caller writes to `detail_path`.  Kept under
    LINE_BUDGET_BYTES whatever the kernel tables hold."""
    line = {k: full.get(k) for k in (
        "metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
        "higher_is_better", "scaling", "vs_baseline", "dtype")}
    line["data"] = _short(full.get("data", "synthetic"), 160)
    cfg = dict(full.get("config") or {})
    if "workload" in cfg:
        cfg["workload"] = _short(cfg["workload"], 200)
    if "parallelism" in cfg:
        cfg["parallelism"] = _short(cfg["parallelism"], 96)
    line["config"] = cfg
    line["window_s"] = full.get("window_s")
    line["timed_steps"] = full.get("timed_steps")
    if full.get("short_window"):
        sw = full["short_window"]
        line["short_window"] = {"steps": sw["steps"], "ms_per_step": sw["ms_per_step"],
                                "value": sw["value"]}
    if full.get("per_rank_ms_per_step") and len(full["per_rank_ms_per_step"]) > 1:
        line["per_rank_ms_per_step"] = full["per_rank_ms_per_step"][:8]
    if full.get("allgather"):
        line["allgather"] = full["allgather"]
    r = full.get("roofline")
    if r:
        nodes = r.get("nodes") or {}
        line["roofline"] = {
            "kernel": _short(r.get("kernel", ""), 80),
            "bound": r.get("bound"), "achieved": r.get("achieved"),
            "peak": r.get("peak"), "unit": r.get("unit"), "frac": r.get("frac"),
            "traffic": r.get("traffic"), "traffic_source": r.get("traffic_source"),
            "avg_us": r.get("avg_us"),
            "algo_bytes_per_launch": r.get("algo_bytes_per_launch"),
            "nodes": {k: _five(nodes.get(k)) for k in (
                "sort_node", "parallel_for", "sort_and_parallel_for",
                "physics_step_issue", "step") if nodes.get(k)},
        }
        pm = r.get("peak_measured")
        if pm:
            line["roofline"]["peak_measured_GBps"] = {
                "copy": pm.get("copy_GBps"), "triad": pm.get("triad_GBps")}
    else:
        line["roofline"] = None
    c = full.get("cpu_baseline")
    if c:
        line["cpu_baseline"] = {"value": c.get("value"), "unit": c.get("unit"),
                                "cores": c.get("cores"), "kind": c.get("kind"),
                                "sample": _short(c.get("sample", ""), 200)}
    else:
        line["cpu_baseline"] = None
    s = full.get("ecs_config2")
    if s:
        sr = s.get("roofline") or {}
        line["ecs_config2"] = {
            "value": s.get("value"), "ms_per_step": s.get("ms_per_step"),
            "sort_node": _five(sr),
            "sort_and_parallel_for": _five((sr.get("nodes") or {})
                                           .get("sort_and_parallel_for"))}
    rd = full.get("render_config5")
    if rd:
        cb = rd.get("cpu_baseline_render_pass") or {}
        line["render_config5"] = {
            "value": rd.get("value"), "ms_per_step": rd.get("ms_per_step"),
            "render_graph_us": rd.get("render_graph_us"),
            "raycast": _five(rd.get("roofline")),
            "cpu_views_per_s": cb.get("value"), "cpu_cores": cb.get("cores")}
    ps = full.get("portable_sim")
    if ps:
        line["portable_sim"] = {k: {"value": v.get("value"),
                                    "ms_per_step": v.get("ms_per_step")}
                                for k, v in ps.items() if isinstance(v, dict)}
    hs = full.get("hideseek_config4_share")
    if hs:
        hr = hs.get("roofline") or {}
        line["hideseek_config4_share"] = {
            "value": hs.get("value"), "ms_per_step": hs.get("ms_per_step"),
            "worlds": 8192, "physics_step_us": hr.get("avg_us")}
    cp = full.get("cartpole_config1")
    if cp:
        line["cartpole_config1"] = {
            "value": cp.get("value"), "ms_per_step": cp.get("ms_per_step"),
            "worlds": 64,
            "cpu_reference": (cp.get("cpu_reference") or {}).get("value")}
    line["detail"] = detail_path
    # whatever happens upstream, the line stays parseable by the driver
    for drop in ("cartpole_config1", "hideseek_config4_share", "portable_sim",
                 "render_config5", "ecs_config2", "short_window", "allgather"):
        if len(json.dumps(line)) < LINE_BUDGET_BYTES:
            break
        line.pop(drop, None)
    assert len(json.dumps(line)) < LINE_BUDGET_BYTES
    return line


def write_detail(full):
    """Per-kernel tables, chains and notes of the run: gpurun_out/ when it exists
    (it travels back from the GPU box), else next to bench.py.  One file per
    workload (bench_detail_<sim>_w<worlds>_g<gpus>.json)."""
    d = os.path.join(REPO, "gpurun_out")
    cfg = full.get("config") or {}
    name = "bench_detail_%s_w%s_g%s.json" % (cfg.get("sim", "sim"),
                                             cfg.get("worlds_per_gpu", 0),
                                             full.get("n_gpus", 1))
    path = os.path.join(d if os.path.isdir(d) else REPO, name)
    try:
        with open(path, "w") as f:
            json.dump(full, f, indent=1)
    except OSError:
        return None
    return os.path.relpath(path, REPO)


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=600)
    p.add_argument("--warmup", type=int, default=100)
    p.add_argument("--worlds", type=int, default=0,
                   help="worlds per GPU (default: the BASELINE size of the workload)")
    p.add_argument("--sim", default="",
                   help="default: escape_room_phys (configs[2]) for every --gpus N "
                        "(one workload along a scaling curve); hideseek = "
                        "configs[3]'s worlds")
    p.add_argument("--dry-launch", action="store_true",
                   help="launch / rendezvous check without a GPU: the ranks meet "
                        "over gloo, all-gather their world ranges and rank 0 prints "
                        "one JSON line (tests/test_bench_contract.py)")
    p.add_argument("--auto-reset-denom", type=int, default=200)
    p.add_argument("--action-slots", type=int, default=61,
                   help="action sets in the device-resident ring the step graph "
                        "takes its actions from, one per step (default 61; 1 = the "
                        "same actions every step; 0 = what rounds 1-2 measured: the "
                        "action tensor written ONCE before the run -- the simulators "
                        "zero a world's actions when it resets, so most agents stand "
                        "still by the time the clock starts)")
    p.add_argument("--settle", type=int, default=-1,
                   help="steps the synthetic worlds are advanced while they are set "
                        "up, before warm-up and timing (default 400: freshly spawned "
                        "piles are still falling and the first hundreds of steps "
                        "carry 15-40 %% more contacts than the steady state the "
                        "metric is about; part of the synthetic-data preparation, "
                        "not of the warm-up)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-secondary", action="store_true",
                   help="skip the configs[1] (physics off, 4096 worlds) measurement")
    p.add_argument("--profile-reps", type=int, default=30)
    p.add_argument("--force-collective", action="store_true",
                   help="run the RCCL observation all-gather even with one rank "
                        "(exercises the multi-GPU code path on a 1-GPU box)")
    return p.parse_args()


def measure_hbm_bandwidth(seconds=0.2):
    """the 8 TB/s spec next to what THIS box's HBM delivers: a device
    copy (read + write, 512 MiB each way per launch: far beyond the caches) and a
    triad a = b + s * c on fp32 (2 reads + 1 write), timed with events on the
    current stream for about `seconds` each.  GB/s of bytes moved."""
    import torch
    n = 128 * 1024 * 1024           # 512 MiB of fp32 per array
    a = torch.empty(n, dtype=torch.float32, device="cuda")
    b = torch.ones(n, dtype=torch.float32, device="cuda")
    c = torch.ones(n, dtype=torch.float32, device="cuda")
    out = {}
    for name, fn, nbytes in (("copy", lambda: a.copy_(b), 2 * 4 * n),
                             ("triad", lambda: torch.add(b, c, alpha=2.0, out=a),
                              3 * 4 * n)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        reps, total_ms = 0, 0.0
        while total_ms < seconds * 1e3 and reps < 2000:
            ev0.record()
            for _ in range(10):
                fn()
            ev1.record()
            torch.cuda.synchronize()
            total_ms += ev0.elapsed_time(ev1)
            reps += 10
        out[name] = nbytes * reps / (total_ms * 1e-3) / 1e9
    del a, b, c
    torch.cuda.empty_cache()
    return {"copy_GBps": round(out["copy"], 1), "triad_GBps": round(out["triad"], 1),
            "note": "torch device copy / fused add on 512 MiB fp32 arrays, events "
                    "around 10 launches at a time"}


def recorded_traffic():
    """HBM bytes per launch measured with the PMC counters (rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE in separate passes; rocprofv3 cannot run inside the
    bench): the newest profiles/rNN_hbm_traffic.json."""
    import glob
    entries = []
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_hbm_traffic.json"))):
        try:
            with open(path) as f:
                for e in json.load(f)["entries"]:
                    e["source"] = os.path.relpath(path, REPO)
                    entries.append(e)
        except (OSError, ValueError, KeyError):
            pass
    return entries


def recorded_issue_counters():
    """SQ instruction / cycle counters per launch (rocprofv3 --pmc, kernel trace
    only, two passes: profiles/tools/refresh_r03.sh + make_issue_json.py): the
    newest profiles/rNN_issue_counters.json."""
    import glob
    entries = []
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_issue_counters.json"))):
        try:
            with open(path) as f:
                for e in json.load(f)["entries"]:
                    e["source"] = os.path.relpath(path, REPO)
                    entries.append(e)
        except (OSError, ValueError, KeyError):
            pass
    return entries


def issue_roofline(name, sim, worlds, kernel_pattern, avg_us, waves_per_simd, note):
    """`valu-issue` roofline of a kernel: wave-instructions issued per cycle and
    SIMD (recorded SQ_INSTS_* per launch / the LIVE event-timed duration of this
    run), against what the waves this kernel keeps per SIMD can issue."""
    match = None
    for e in recorded_issue_counters():
        if (e["sim"], e["worlds"]) == (sim, worlds) and kernel_pattern in e["kernel"]:
            # later files override earlier ones; of one file's entries the one
            # that issued the most (the fallback launch behind the LDS step --
            # "physics:worldStep(HBM)" -- matches the same pattern and is empty)
            if (match is None or e["source"] != match["source"] or
                    e.get("SQ_INSTS_VALU", 0.0) > match.get("SQ_INSTS_VALU", 0.0)):
                match = e
    if match is None or avg_us <= 0 or "SQ_INSTS_VALU" not in match:
        return None
    insts = (match.get("SQ_INSTS_VALU", 0.0) + match.get("SQ_INSTS_SALU", 0.0) +
             match.get("SQ_INSTS_LDS", 0.0))
    cycles = avg_us * 1e-6 * SHADER_CLOCK_HZ
    achieved = insts / (cycles * NUM_SIMDS)
    peak = ISSUE_PEAK_ONE_WAVE if waves_per_simd <= 1 else ISSUE_PEAK_SIMD
    out = {
        "kernel": name, "bound": "valu-issue",
        "achieved": round(achieved, 4), "peak": peak,
        "unit": "wave-instructions/cycle/SIMD", "frac": round(achieved / peak, 4),
        "valu_only": round(match["SQ_INSTS_VALU"] / (cycles * NUM_SIMDS), 4),
        "waves_per_simd": waves_per_simd, "avg_us": round(avg_us, 2),
        "insts_per_launch": {k: match[k] for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU",
                                                   "SQ_INSTS_LDS", "SQ_WAVES")
                             if k in match},
        "counters_source": match["source"], "note": note,
    }
    wc = match.get("SQ_WAVE_CYCLES")
    if wc:
        out["wave_cycles"] = {
            "issuing": round(match.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 3),
            "parked_on_waitcnt": round(match.get("SQ_WAIT_ANY", 0.0) / wc, 3),
            "issue_stalls": round(match.get("SQ_WAIT_INST_ANY", 0.0) / wc, 3),
        }
    return out


def traffic_for(entries, sim, worlds, kernel_pattern, per_launch=False):
    """Sum of the newest recorded traffic of every kernel whose name contains
    `kernel_pattern` for this workload, or None.  Per replay of the recorded run,
    or (per_launch) per launch of the kernel -- config 5 replays two graphs per
    environment step, so its ray caster runs in every other replay."""
    latest = {}
    for e in entries:
        if (e["sim"], e["worlds"]) == (sim, worlds) and kernel_pattern in e["kernel"]:
            latest[e["kernel"]] = e     # later files override earlier ones
    if not latest:
        return None, None

    def amount(e):
        if per_launch and e.get("launches_per_step"):
            return e["traffic_bytes"] / e["launches_per_step"]
        return e["traffic_bytes"]
    return (int(sum(amount(e) for e in latest.values())),
            sorted({e["source"] for e in latest.values()})[-1])


def cpu_baseline(sim, worlds, flags, seed, settle, budget_s=12.0):
    """The reference's own CPU backend (oracle/_ref, speed build) on this box's
    host cores, bounded sample of the same workload.  Reported, not a target."""
    from madrona_amd.simlib import Simulator, ref_lib_path
    path = ref_lib_path(sim, speed=True)
    if not os.path.exists(path):
        return None
    cores = len(os.sched_getaffinity(0))
    with Simulator(path, worlds, seed=seed, num_workers=0, flags=flags) as s:
        # (same preparation as the GPU run, capped: the CPU needs ~60 ms a step)
        settle = min(settle, 150)
        import numpy as np
        rng = np.random.default_rng(seed)
        ring = [np.stack(random_actions(sim, worlds, lambda lo, hi, shape:
                                        rng.integers(lo, hi, shape)), -1).astype(np.int32)
                for _ in range(max(ACTION_SLOTS, 1))]
        count = [0]

        def step(n):
            # a new action set every step, like the GPU run's input ring
            for _ in range(n):
                if ACTION_SLOTS != 0 or count[0] == 0:
                    s.write_tensor("action", ring[count[0] % max(ACTION_SLOTS, 1)])
                count[0] += 1
                s.step(1)

        step(settle)
        step(5)
        t0 = time.perf_counter()
        step(5)
        per_step = (time.perf_counter() - t0) / 5
        n = int(max(20, min(20000, budget_s / max(per_step, 1e-6))))
        t0 = time.perf_counter()
        step(n)
        dt = time.perf_counter() - t0
    return {
        "value": worlds * n / dt, "unit": "steps/s", "cores": cores,
        "kind": "reference",
        "sample": f"{sim}, {worlds} worlds x {n} steps ({dt:.1f} s) after {settle} "
                  f"settling steps, on the reference TaskGraphExecutor (oracle/_ref "
                  f"speed build, numWorkers=0)",
    }


ACTION_SLOTS = 61   # (odd: config 5 replays two graphs per step); --action-slots
ACTION_WORKLOAD = ("a new set of random actions every step from a 61-slot ring "
                   "resident in HBM")


def random_actions(sim_name, worlds, rng_randint):
    """One set of policy outputs: move amount / angle / rotate / grab per agent."""
    A = AGENTS.get(sim_name, 2)
    cols = [rng_randint(0, 4, (worlds, A)), rng_randint(0, 8, (worlds, A)),
            rng_randint(-2, 3, (worlds, A)), rng_randint(0, 2, (worlds, A))]
    if sim_name == "escape_room":
        cols[3] = cols[3] * 0       # (no grab action without physics)
    return cols


def fill_actions(sim_name, sim, worlds, gpu_id, seed):
    """Synthetic policy outputs, resident in HBM before any timed region: a ring
    of ACTION_SLOTS action sets; every replay of the step graph starts by copying
    the next one into the exported action tensor (mwhip_set_input_ring), so the
    worlds are driven by fresh actions every step without the host touching the
    executor's stream.  (Rounds 1-2 wrote the tensor once: the simulators zero a
    world's actions when it resets, so by the end of the settling steps most
    agents stood still -- no grabs, an idle joint sort.  --action-slots 0.)"""
    import torch
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    ring = torch.stack([
        torch.stack(random_actions(sim_name, worlds, lambda lo, hi, shape:
                                   torch.randint(lo, hi, shape, device="cuda",
                                                 generator=gen)), -1)
        for _ in range(max(ACTION_SLOTS, 1))]).to(torch.int32).contiguous()
    torch.cuda.synchronize()
    if ACTION_SLOTS == 0:
        # (--action-slots 0: rounds 1-2)
        from madrona_amd.tensor import to_torch
        to_torch(sim, "action", gpu_id).copy_(ring[0])
        torch.cuda.synchronize()
        return
    sim._action_ring = ring         # (keeps the memory alive)
    sim.set_input_ring("action", ring.data_ptr(), ACTION_SLOTS)


def annotate(stats, sim_name, worlds):
    """Algorithmic bytes the executor cannot know: the fused physics step."""
    for k in stats:
        if k["name"] == "physics:worldStep(fallback)":
            # (the worlds the LDS step could not hold: usually none)
            k["algo_bytes"] = 0.0
            k["exact"] = False
        elif k["name"].startswith("physics:worldStep"):
            k["algo_bytes"] = (float(worlds) * PHYS_BYTES_PER_BODY *
                               PHYS_BODIES.get(sim_name, 28))
            k["exact"] = True
        elif k["name"] in ("physics:bvhRefresh", "physics:bvhRefresh+rebuild"):
            # the leaf update + refit of every body, a wavefront per world: the
            # declared read / write set of the ParallelFor node it replaces
            # (systemIO<updateLeafAndRefitEntry>, physics.inl) x the body rows
            k["algo_bytes"] = (float(worlds) * PHYS_BODIES.get(sim_name, 28) *
                               LEAF_REFRESH_BYTES_PER_BODY)
            k["rows"] = float(worlds) * PHYS_BODIES.get(sim_name, 28)
            k["io_declared"] = True
            k["exact"] = True
        else:
            k["exact"] = ":sort." in k["name"]
    return stats


def kernel_table(stats, sim_name, worlds):
    """mwhip_profile output -> JSON rows.  ParallelFor bytes come from the
    system's signature (`T&` = read + write): an upper bound on what it moves
    (rows that return early move less), so no GB/s is derived from them."""
    rows = []
    for k in annotate(stats, sim_name, worlds):
        row = {"name": k["name"], "avg_us": round(k["avg_us"], 2),
               "rows": round(k["rows"], 1)}
        if k["exact"]:
            row["algo_MB"] = round(k["algo_bytes"] / 1e6, 4)
            row["GBps"] = (round(k["algo_bytes"] / (k["avg_us"] * 1e-6) / 1e9, 1)
                           if k["avg_us"] > 0 else 0.0)
        elif k["algo_bytes"] > 0 and k.get("io_declared"):
            # rows x (4 + declared reads + declared writes), SURVEY 8d
            row["algo_MB"] = round(k["algo_bytes"] / 1e6, 4)
            row["GBps"] = (round(k["algo_bytes"] / (k["avg_us"] * 1e-6) / 1e9, 1)
                           if k["avg_us"] > 0 else 0.0)
            row["bytes"] = "declared read/write set"
        elif k["algo_bytes"] > 0:
            row["algo_MB_signature_upper_bound"] = round(k["algo_bytes"] / 1e6, 4)
        rows.append(row)
    return rows


def node_roofline(name, kernels, traffic, traffic_source, note):
    """One roofline entry over a group of kernels: summed algorithmic bytes /
    summed average durations (HIP events attached to the dispatches)."""
    if not kernels:
        return None
    algo = sum(k["algo_bytes"] for k in kernels)
    us = sum(k["avg_us"] for k in kernels)
    achieved = algo / (us * 1e-6) / 1e9 if us > 0 else 0.0
    return {
        "kernel": name, "bound": "hbm",
        "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4),
        "traffic": traffic, "traffic_source": traffic_source,
        "avg_us": round(us, 2), "launches": len(kernels),
        "algo_bytes_per_launch": int(algo),
        "note": note,
    }


def rooflines(stats, sim_name, worlds, ms_per_step):
    """`roofline` = the kernel that dominates the step's time (contract), plus
    `roofline.nodes`: the WHOLE sort node (every kernel of every sort chain in
    the step), the fused physics step, and the step as a whole."""
    entries = recorded_traffic()
    annotate(stats, sim_name, worlds)
    sort_k = [k for k in stats if ":sort." in k["name"]]
    # (the fallback launch behind the LDS step -- usually empty -- is not part
    # of the dominant kernel's figures)
    phys_k = [k for k in stats if k["name"].startswith("physics:worldStep")
              and k["name"] != "physics:worldStep(fallback)"]

    nodes = {}
    t, src = traffic_for(entries, sim_name, worlds, ":sort.")
    sort_roles = sorted({k["name"].split(":sort.", 1)[1] for k in sort_k})
    nodes["sort_node"] = node_roofline(
        "SortArchetype/CompactArchetype nodes: every kernel of every sort chain in "
        "the step (" + ", ".join(sort_roles) + ")", sort_k, t, src,
        "BASELINE's 'achieved HBM GB/s on sort node': all kernels of the node, not "
        "its best one; `frac` prices the node at SURVEY 8d's formula x the rows the "
        "sorts measured (40 N + 2 B_row N': what any implementation of the node is "
        "priced at); `frac_bytes_moved` at what these kernels move -- world sorts "
        "of tables that are still sorted from the last step take the compaction "
        "chain (prepare + scatter: 8 N + 8 N' instead of the 40 N of histogram + "
        "key passes), so 8 N + 8 N' + 2 B_row N'")
    if nodes["sort_node"]:
        # the step's sort chains one by one (kernels of a chain are adjacent in
        # the launch list): e.g. the joint table before the physics step -- a
        # few thousand rows behind three launch floors -- next to the compaction
        # of the big tables after the reset
        chains, run = [], []
        for k in stats:
            if ":sort." in k["name"]:
                run.append(k)
            elif run:
                chains.append(run)
                run = []
        if run:
            chains.append(run)
        nodes["sort_node"]["chains"] = [{
            "kernels": [k["name"].split(":", 1)[1] for k in c],
            "avg_us": round(sum(k["avg_us"] for k in c), 2),
            "algo_bytes": int(sum(k["algo_bytes"] for k in c)),
            "frac": round(sum(k["algo_bytes"] for k in c) /
                          max(sum(k["avg_us"] for k in c) * 1e-6, 1e-12) / 1e9 /
                          HBM_PEAK_GBS, 4),
        } for c in chains]
        rows_in = sum(k["rows"] for k in sort_k if "gather" in k["name"])
        gather_bytes = sum(k["algo_bytes"] for k in sort_k if "gather" in k["name"])
        # What the chains MOVE (next to what SURVEY 8d prices a sort node at):
        # a compaction chain reads 4 N keys + writes 4 N' keys + 4 N' indices
        # and re-reads them in the gather (8 N + 8 N') instead of the 40 N of
        # histogram + key passes; the gather is 2 B_row N' either way.
        compact_rows = sum(k["rows"] for k in sort_k if "compact.prepare" in k["name"])
        radix_bytes = sum(k["algo_bytes"] for k in sort_k
                          if "histogram" in k["name"] or "onesweep" in k["name"]
                          or "small" in k["name"])
        moved = gather_bytes + 16.0 * compact_rows + radix_bytes
        node_us = nodes["sort_node"]["avg_us"]
        nodes["sort_node"]["bytes_moved_estimate"] = int(moved)
        nodes["sort_node"]["achieved_bytes_moved"] = round(
            moved / max(node_us * 1e-6, 1e-12) / 1e9, 1)
        nodes["sort_node"]["frac_bytes_moved"] = round(
            moved / max(node_us * 1e-6, 1e-12) / 1e9 / HBM_PEAK_GBS, 4)
    t, src = traffic_for(entries, sim_name, worlds, "physics:worldStep")
    nodes["physics_step"] = node_roofline(
        "physics:worldStep (broadphase pairs + 4 x (integrate, narrowphase, XPBD "
        "position + velocity solve) fused, two worlds per wavefront, worlds in LDS)",
        phys_k, t, src,
        "latency/issue-bound: HBM sees one read + one write of the body columns "
        "per step (288 B/body), loaded through a per-launch frame of addresses in "
        "six rounds of global loads; DESIGN.md §10, §14.9")
    if phys_k:
        nodes["physics_step_issue"] = issue_roofline(
            "physics:worldStep (same kernel, instruction-issue yardstick)", sim_name,
            worlds, "physics:worldStep", sum(k["avg_us"] for k in phys_k),
            2,
            "two worlds per wavefront, two wavefronts per SIMD since round 6 (LDS: "
            "20 256 B per workgroup, 256 registers per wavefront): the kernel is "
            "bound by instruction issue and LDS / scratch latency, not by HBM "
            "(DESIGN.md §10, §16)")
    # ParallelFor nodes (north_star: "sort + ParallelFor nodes at >= 50 % of the
    # HBM roofline"): bytes = rows x (4 + declared reads + declared writes) where
    # the system declares its read / write set next to its definition
    # (madrona::mwhip::systemIO), else the signature rule (an upper
    # bound: T & = read + write), kept apart.
    pfor_k = [k for k in stats if k["algo_bytes"] > 0 and ":sort." not in k["name"]
              and not k["name"].startswith("physics:worldStep")]
    for label, group in (("parallel_for", [k for k in pfor_k if k.get("io_declared")]),
                         ("parallel_for_signature_rule",
                          [k for k in pfor_k if not k.get("io_declared")])):
        if not group:
            continue
        t_sum, t_src, t_missing, seen = 0, None, [], set()
        for k in group:
            # (nodes sharing a launch run in pforGroupKernel: the counters see
            # ONE kernel for all such launches of a step)
            key = "group[" if k["name"].startswith("group[") else k["name"]
            if key in seen:
                continue        # (recorded traffic is per STEP: every launch of
            seen.add(key)           # a kernel that runs several times is in it)
            t, src = traffic_for(entries, sim_name, worlds, key)
            if t is None:
                t_missing.append(k["name"])
            else:
                t_sum, t_src = t_sum + t, src
        node = node_roofline(
            "ParallelFor nodes, read / write sets declared next to the systems"
            if label == "parallel_for" else
            "ParallelFor nodes without a declared set (signature rule: upper bound)",
            group, t_sum if t_src else None, t_src,
            "sum of rows x (4 B WorldID + declared reads + declared writes) over "
            "the nodes / sum of their kernel times (HIP events on the dispatches); "
            "at Escape-Room sizes most of these launches sit on the ~4.5 us floor "
            "of a graph kernel node, not on bytes")
        node["kernels"] = [{
            "name": k["name"], "avg_us": round(k["avg_us"], 2),
            "rows": round(k["rows"], 1), "algo_MB": round(k["algo_bytes"] / 1e6, 4),
            "GBps": round(k["algo_bytes"] / max(k["avg_us"] * 1e-6, 1e-12) / 1e9, 1),
        } for k in group]
        if t_missing:
            node["traffic_missing_for"] = t_missing
        nodes[label] = node
    # sort + declared ParallelFor nodes together: the clause as north_star words it
    both = sort_k + [k for k in pfor_k if k.get("io_declared")]
    if sort_k and len(both) > len(sort_k):
        nodes["sort_and_parallel_for"] = node_roofline(
            "sort nodes + ParallelFor nodes (declared sets) of the step", both,
            None, None, "north_star's clause taken as a whole: summed algorithmic "
            "bytes / summed kernel time")

    # step level: every kernel's algorithmic bytes over the measured step time
    step_bytes = sum(k["algo_bytes"] for k in stats)
    t, src = traffic_for(entries, sim_name, worlds, "step:all-kernels")
    achieved = step_bytes / (ms_per_step * 1e-3) / 1e9
    nodes["step"] = {
        "kernel": "whole step (all launches of one replay)", "bound": "hbm",
        "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4),
        "traffic": t, "traffic_source": src,
        "algo_bytes_per_step": int(step_bytes),
        "note": "sum of algorithmic bytes (ParallelFor nodes at their signature "
                "upper bound) / ms_per_step of the timed window",
    }
    nodes = {k: v for k, v in nodes.items() if v}

    dominant = None
    if phys_k and sum(k["avg_us"] for k in phys_k) >= sum(k["avg_us"] for k in sort_k):
        dominant = dict(nodes["physics_step"])
    elif sort_k:
        dominant = dict(nodes["sort_node"])
    if dominant is not None:
        dominant["nodes"] = nodes
        dominant["event_floor_us"] = round(min(k["avg_us"] for k in stats), 2)
    return dominant


def run_single(sim_name, worlds, gpu_id, seed, denom, steps, warmup, profile_reps,
               settle=0):
    """One-GPU measurement of `sim_name` outside the distributed harness (the
    secondary configs[1] line).  Same timing discipline as the headline."""
    import torch
    from madrona_amd.simlib import Simulator, hip_lib_path

    # "<sim>_portable": the same simulator built from its portable sources
    lib_name, sim_name = sim_name, sim_name.removesuffix("_portable")
    with Simulator(hip_lib_path(lib_name), worlds, seed=seed, gpu_id=gpu_id,
                   flags=denom) as sim:
        fill_actions(sim_name, sim, worlds, gpu_id, 99)
        sim.step_async(settle + warmup)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sim.step_async(steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        sim.sync()
        stats = sim.profile(profile_reps)
    ms = dt / steps * 1e3
    return {
        "workload": WORKLOADS[sim_name][1].format(w=worlds) +
                    f", auto-reset p=1/{denom} per world per step",
        "value": worlds * steps / dt, "unit": "steps/s", "ms_per_step": ms,
        "steps": steps, "warmup": warmup,
        "roofline": rooflines(stats, sim_name, worlds, ms),
        "kernels": kernel_table(stats, sim_name, worlds),
    }


def run_cartpole(gpu_id, worlds=64, steps=2000, cpu_sample=True):
    """BASELINE configs[0]: the Cartpole-style 2-component ECS, 64 worlds -- quoted
    on the reference's CPU executor (plumbing); the same simulator on the HIP
    executor next to it.  At 64 worlds a step is launch floors, not bandwidth."""
    import numpy as np
    import torch
    from madrona_amd.simlib import Simulator, hip_lib_path, ref_lib_path
    rng = np.random.default_rng(3)
    actions = rng.integers(0, 2, (worlds, 1)).astype(np.int32)
    out = {"workload": f"Cartpole-style 2-component ECS, {worlds} worlds "
                       "(BASELINE.json configs[0])"}
    with Simulator(hip_lib_path("cartpole"), worlds, seed=5, gpu_id=gpu_id) as sim:
        sim.write_tensor("action", actions)
        sim.step_async(200)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sim.step_async(steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        sim.sync()
    out.update(value=worlds * steps / dt, unit="steps/s", ms_per_step=dt / steps * 1e3)
    path = ref_lib_path("cartpole", speed=True)
    if not os.path.exists(path):
        path = ref_lib_path("cartpole")
    if cpu_sample and os.path.exists(path):
        with Simulator(path, worlds, seed=5, num_workers=0) as ref:
            ref.write_tensor("action", actions)
            ref.step(200)
            t0 = time.perf_counter()
            ref.step(steps)
            dt = time.perf_counter() - t0
        out["cpu_reference"] = {"value": worlds * steps / dt, "unit": "steps/s",
                                "cores": len(os.sched_getaffinity(0)),
                                "ms_per_step": dt / steps * 1e3}
    return out


def cpu_raycast_sample(sim, worlds, resolution, sample_worlds=8192):
    """cpu_baseline leg of config 5's render pass: the reference's own ray caster
    (src/mw/device/bvh_raycast.cpp compiled for the host, oracle/_ref/
    libraycast_ref.so) over the instance / view / light rows of the first
    `sample_worlds` worlds, one thread per host core.  Its QBVHs are built by the
    oracle shim (median splits), not by Embree: a reported baseline only."""
    import ctypes as C
    import numpy as np
    path = os.path.join(REPO, "oracle", "_ref", "libraycast_ref.so")
    if not os.path.exists(path):
        return None
    lib = C.CDLL(path)
    p = C.c_void_p
    lib.raycast_ref_render.restype = C.c_int
    lib.raycast_ref_render.argtypes = [C.c_uint32, p, p, p, p, p, C.c_uint32, p,
                                       C.c_uint32, p, p, p, p, C.c_uint32, p, p, p,
                                       C.c_uint32, C.c_uint32, C.c_uint32, p, p,
                                       p, p, p, C.c_uint32, p, p]
    geo = sim.lib.sim_render_geometry
    geo.restype = C.c_int32
    geo.argtypes = [p] * 7
    counts = np.zeros(3, np.uint32)
    n_obj = geo(None, None, None, None, None, None, counts.ctypes.data)
    verts = np.zeros((counts[0], 3), np.float32)
    idx = np.zeros((counts[1], 3), np.uint32)
    voff = np.zeros(n_obj + 1, np.uint32)
    toff = np.zeros(n_obj + 1, np.uint32)
    mats = np.zeros((counts[2], 3), np.float32)
    omat = np.zeros(n_obj, np.int32)
    geo(verts.ctypes.data, idx.ctypes.data, voff.ctypes.data, toff.ctypes.data,
        mats.ctypes.data, omat.ctypes.data, None)

    d = sim.dump_all()
    inst, ic = d["Renderable.InstanceData"]
    views, vc = d["Camera.PerspectiveCameraData"]
    lights, lc = d["Light.LightDesc"]
    W = min(sample_worlds, worlds)
    ic, vc, lc = ic[:W], vc[:W], lc[:W]
    offs = lambda c: np.concatenate([[0], np.cumsum(c)[:-1]]).astype(np.int32)
    nv = int(vc.sum())
    inst = np.ascontiguousarray(inst[:int(ic.sum())])
    views = np.ascontiguousarray(views[:nv])
    lights = np.ascontiguousarray(lights[:max(int(lc.sum()), 1)])
    io, lo = offs(ic), offs(lc)
    rgb = np.zeros((nv, resolution, resolution, 4), np.uint8)
    depth = np.zeros((nv, resolution, resolution), np.float32)
    cores = os.cpu_count() or 1
    t0 = time.perf_counter()
    rc = lib.raycast_ref_render(
        n_obj, verts.ctypes.data, idx.ctypes.data, voff.ctypes.data, toff.ctypes.data,
        omat.ctypes.data, len(mats), mats.ctypes.data, W, inst.ctypes.data,
        io.ctypes.data, np.ascontiguousarray(ic, np.int32).ctypes.data,
        views.ctypes.data, nv, lights.ctypes.data, lo.ctypes.data,
        np.ascontiguousarray(lc, np.int32).ctypes.data, resolution, 1,
        min(cores, resolution), rgb.ctypes.data, depth.ctypes.data,
        None, None, None, 0, None, None)    # (the Escape Room's meshes are untextured)
    dt = time.perf_counter() - t0
    if rc != 0:
        return None
    return {"value": nv / dt, "unit": "views/s", "cores": min(cores, resolution),
            "kind": "reference",
            "sample": f"{nv} views of {resolution}x{resolution} RGB-D ({W} worlds, "
                      f"{dt:.2f} s incl. building its BVHs) through the reference's "
                      f"bvhRaycastEntry compiled for the host (oracle/ref_shims/"
                      f"raycast_ref_shim.cpp)"}


def run_render(worlds, gpu_id, seed, denom, steps, warmup, profile_reps, settle,
               resolution=64, cpu_sample=True):
    """BASELINE.json configs[4]: the physics Escape Room + the batch ray caster,
    resolution^2 RGB-D per agent.  A step = one replay of the step graph followed
    by one replay of the render graph (TLAS build + ray cast of every view)."""
    import torch
    from madrona_amd.simlib import Simulator, hip_lib_path

    flags = denom | (resolution << 16)
    with Simulator(hip_lib_path("escape_room_render"), worlds, seed=seed, gpu_id=gpu_id,
                   flags=flags) as sim:
        fill_actions("escape_room_phys", sim, worlds, gpu_id, 77)
        sim.step_async(settle)
        render = sim.render_graph()

        def run(n):
            for _ in range(n):
                sim.step_async(1)
                sim.step_async(1, graph=render)

        from madrona_amd.simlib import runtime_lib
        rt = runtime_lib()
        run(warmup)
        torch.cuda.synchronize()
        # marker dispatches: the rocprofv3 summaries in profiles/ are trimmed to
        # the kernels between them (profiles/summarize_rocprof.py)
        rt.mwhip_mark_window(sim.hip_exec(), 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rt.mwhip_mark_window(sim.hip_exec(), 2)
        sim.sync()
        step_stats = sim.profile(profile_reps)
        render_stats = sim.profile(profile_reps, graph=render)
        cpu_views = cpu_raycast_sample(sim, worlds, resolution) if cpu_sample else None
        views = worlds * AGENTS["escape_room_phys"]
        # instances per world: floor, 4 borders, 2 agents, 3 x (2 walls, door, 4
        # cubes, 2 buttons)
        instances = 34
    ms = dt / steps * 1e3
    rays = views * resolution * resolution
    cast = [k for k in render_stats if "raycast" in k["name"]]
    tlas = [k for k in render_stats if "tlas" in k["name"]]
    cast_us = sum(k["avg_us"] for k in cast)
    # what a view has to move: its world's instance records, leaf boxes and TLAS
    # nodes in, resolution^2 x (RGBA8 + f32 depth) out
    cast_bytes = views * (instances * (64 + 32 + 64) + resolution * resolution * 8)
    cast_traffic, cast_traffic_src = traffic_for(
        recorded_traffic(), "escape_room_render", worlds, "render:raycast",
        per_launch=True)
    out = {
        "workload": f"Escape-Room + XPBD + batch ray caster, {resolution}x{resolution} "
                    f"RGB-D per agent, {worlds} worlds (BASELINE.json configs[4]), "
                    f"{views} views, {instances} instances/world, 1 directional light, "
                    f"auto-reset p=1/{denom} per world per step",
        "value": worlds * steps / dt, "unit": "steps/s", "ms_per_step": ms,
        "steps": steps, "warmup": warmup,
        "step_graph_us": round(sum(k["avg_us"] for k in step_stats), 1),
        "render_graph_us": round(sum(k["avg_us"] for k in render_stats), 1),
        "primary_rays_per_s": rays / (cast_us * 1e-6) if cast_us > 0 else None,
        "views_per_s_render_pass": views / (sum(k["avg_us"] for k in render_stats) * 1e-6),
        "cpu_baseline_render_pass": cpu_views,
        "roofline": {
            "kernel": "render:raycast (16x16 tile of one view per workgroup, world's "
                      "TLAS + instances in LDS, binary BVH traversal, shading)",
            "bound": "hbm", "achieved": round(cast_bytes / (cast_us * 1e-6) / 1e9, 1)
            if cast_us > 0 else 0.0,
            "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(cast_bytes / (cast_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
            if cast_us > 0 else 0.0,
            "traffic": cast_traffic, "traffic_source": cast_traffic_src,
            "avg_us": round(cast_us, 1),
            "algo_bytes_per_launch": int(cast_bytes),
            "note": "traversal is instruction/latency bound: the HBM figure says how "
                    "far the kernel is from merely writing its images",
        },
        "kernels": [{"name": k["name"], "avg_us": round(k["avg_us"], 2)}
                    for k in render_stats],
        "tlas_build_us": round(sum(k["avg_us"] for k in tlas), 2),
    }
    issue = issue_roofline(
        "render:raycast (same kernel, instruction-issue yardstick)",
        "escape_room_render", worlds, "render:raycast", cast_us, 3,
        "a traversal: bound by instruction issue and the latency of dependent node "
        "/ triangle fetches from LDS, not by HBM (DESIGN.md §12); three 256-thread "
        "workgroups per CU = three waves per SIMD")
    if issue is not None:
        out["roofline"]["nodes"] = {"raycast_issue": issue}
    return out


def self_launch(gpus):
    """`python bench.py --gpus N` outside torchrun: re-runs this command line as N
    ranks (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1`, a free port) and returns its exit code; the ranks'
    stdout (rank 0's ONE line) and stderr pass through."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def dry_launch(args, rank, world_size, saved_stdout):
    """The launch path without a GPU: every rank joins a gloo group on
    127.0.0.1, the ranks all-gather their world ranges (the sharding the real run
    uses) and rank 0 prints one line."""
    import torch
    import torch.distributed as dist
    from madrona_amd.distributed import shard_for
    if world_size > 1:
        dist.init_process_group("gloo")
    shard = shard_for(rank, world_size, worlds_per_rank=args.worlds or 8192)
    mine = torch.tensor([shard.world_base, shard.worlds_per_rank], dtype=torch.int64)
    ranges = [mine]
    if world_size > 1:
        ranges = [torch.zeros_like(mine) for _ in range(world_size)]
        dist.all_gather(ranges, mine)
        dist.barrier()
    if world_size > 1:
        dist.destroy_process_group()
    import ctypes
    ctypes.CDLL(None).fflush(None)
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    if rank == 0:
        print(json.dumps({
            "dry_launch": True, "n_gpus": args.gpus, "ranks": world_size,
            "backend": "gloo", "sim": args.sim,
            "world_ranges": [[int(r[0]), int(r[1])] for r in ranges],
            "total_worlds": shard.total_worlds}), flush=True)


def main():
    args = parse_args()
    global ACTION_SLOTS, ACTION_WORKLOAD
    ACTION_SLOTS = max(args.action_slots, 0)
    if ACTION_SLOTS == 0:
        ACTION_WORKLOAD = ("the action tensor written ONCE before the run and zeroed by "
                           "every world's first reset -- the workload of rounds 1-2, "
                           "not the default")
    elif ACTION_SLOTS == 1:
        ACTION_WORKLOAD = "the same set of random actions re-applied every step"
    elif ACTION_SLOTS != 61:
        ACTION_WORKLOAD = (f"a new set of random actions every step from a "
                           f"{ACTION_SLOTS}-slot ring resident in HBM")

    # stdout carries exactly one JSON line: anything libraries print while the
    # benchmark runs (RCCL prints a version banner from C) goes to stderr
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` by itself: one rank per GPU under
        # torch.distributed.run, this process only relays rank 0's line
        os.dup2(saved_stdout, 1)
        raise SystemExit(self_launch(args.gpus))
    if world_size != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_size}")
    if not args.sim:
        # ONE workload for every N: the points of a scaling run compare
        args.sim = "escape_room_phys"
    if args.dry_launch:
        return dry_launch(args, rank, world_size, saved_stdout)
    default_worlds, workload_fmt = WORKLOADS.get(args.sim, (4096, args.sim + ", {w} worlds"))
    if args.worlds <= 0:
        args.worlds = default_worlds

    if args.sim == "escape_room_render":
        # BASELINE configs[4] as the line itself (one GPU): step graph + render
        # graph per step
        if world_size != 1:
            raise SystemExit("--sim escape_room_render is a one-GPU line")
        import torch
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the HIP backend has no CPU fallback")
        torch.cuda.set_device(local_rank)
        if args.settle < 0:
            args.settle = 200
        worlds = args.worlds
        hbm_measured = measure_hbm_bandwidth()
        r = run_render(worlds, local_rank, 5, args.auto_reset_denom, args.steps,
                       args.warmup, args.profile_reps, args.settle,
                       cpu_sample=not args.no_cpu_baseline)
        out = {
            "metric": "aggregate env steps/sec", "value": r["value"], "unit": "steps/s",
            "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": r["ms_per_step"], "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": f"synthetic (worlds advanced {args.settle} steps while being set "
                    f"up; {ACTION_WORKLOAD})",
            "config": {"workload": r["workload"], "sim": args.sim,
                       "worlds_per_gpu": worlds, "settle_steps": args.settle,
                       "total_worlds": worlds, "dist_world_size": 1,
                       "parallelism": "worlds sharded over 1 GPU(s)"},
            "roofline": dict(r["roofline"], peak_measured=hbm_measured),
            "cpu_baseline": r["cpu_baseline_render_pass"],
            "render": {k: r[k] for k in ("step_graph_us", "render_graph_us",
                                         "primary_rays_per_s",
                                         "views_per_s_render_pass", "tlas_build_us",
                                         "kernels")},
        }
        import ctypes
        ctypes.CDLL(None).fflush(None)
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(driver_line(out, write_detail(out))), flush=True)
        return

    import torch
    import torch.distributed as dist

    distributed = world_size > 1 or args.force_collective

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP backend has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if distributed:
        # (--force-collective outside torchrun: a one-rank group on this host)
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from madrona_amd.distributed import ShardedSimulator, shard_for
    from madrona_amd.simlib import Simulator, hip_lib_path, runtime_lib

    # the spec peak next to what this box's HBM delivers
    hbm_measured = measure_hbm_bandwidth() if rank == 0 else None

    shard = shard_for(rank, world_size, worlds_per_rank=args.worlds)
    seed = 5

    def make_sim(num_worlds, world_base):
        return Simulator(hip_lib_path(args.sim), num_worlds, seed=seed,
                         gpu_id=local_rank, world_base=world_base,
                         flags=args.auto_reset_denom)

    obs_tensors = OBS_TENSORS.get(args.sim, OBS_TENSORS["escape_room"])
    sharded = ShardedSimulator(make_sim, shard, obs_tensors if distributed else [])
    sim = sharded.sim
    fill_actions(args.sim, sim, args.worlds, local_rank, 1234 + rank)
    rt = runtime_lib()

    # synthetic-data preparation: worlds advanced to their steady state
    if args.settle < 0:
        args.settle = 400
    sim.step_async(args.settle)
    sim.sync()

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(steps, mark):
        """barrier + sync, `steps` replays, barrier + sync; max over ranks."""
        barrier()
        if mark:
            rt.mwhip_mark_window(sim.hip_exec(), 1)
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            sharded.step(1)
        barrier()
        local = time.perf_counter() - t0
        if mark:
            rt.mwhip_mark_window(sim.hip_exec(), 2)
        sharded.sync()      # device-side error flags of the queued replays
        per_rank = [local]
        if distributed:
            t = torch.tensor([local], device="cuda", dtype=torch.float64)
            gathered = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
            dist.all_gather(gathered, t)
            per_rank = [float(g.item()) for g in gathered]
        return max(per_rank), per_rank

    for _ in range(args.warmup):
        sharded.step(1)

    # the contract's window: exactly --steps replays (the rocprofv3 summaries in
    # profiles/ are trimmed to the marker kernels around it)
    elapsed, per_rank = timed(args.steps, mark=True)
    total_worlds = shard.total_worlds
    value = total_worlds * args.steps / elapsed
    ms_per_step = elapsed / args.steps * 1e3

    # A window this short (the driver's --steps 20 is ~25 ms) says little: it is
    # repeated over enough steps for MIN_WINDOW_S, and THAT window is the line's
    # `value` / `ms_per_step`; the --steps window stays next to it as
    # `short_window`.
    long_window = None
    short_window = None
    if elapsed < MIN_WINDOW_S:
        more = int(math.ceil(MIN_WINDOW_S / max(elapsed / args.steps, 1e-7)))
        el2, per_rank2 = timed(more, mark=False)
        long_window = {"steps": more, "seconds": round(el2, 4),
                       "ms_per_step": el2 / more * 1e3,
                       "value": total_worlds * more / el2}
        short_window = {"steps": args.steps, "seconds": round(elapsed, 5),
                        "ms_per_step": ms_per_step, "value": value,
                        "per_rank_ms_per_step":
                            [round(t / args.steps * 1e3, 5) for t in per_rank]}
        value = long_window["value"]
        ms_per_step = long_window["ms_per_step"]
        per_rank_ms = [round(t / more * 1e3, 5) for t in per_rank2]
    else:
        per_rank_ms = [round(t / args.steps * 1e3, 5) for t in per_rank]

    # the collective on its own (packed observation record, RCCL all-gather)
    allgather = None
    if distributed and sharded._packed_global is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        barrier()
        ev0.record()
        for _ in range(reps):
            sharded._exchange(sharded._packed_local[0])
        ev1.record()
        torch.cuda.synchronize()
        nbytes = sharded._packed_global.numel() * 4
        allgather = {"ms": ev0.elapsed_time(ev1) / reps, "gathered_bytes": nbytes,
                     "per_rank_bytes": nbytes // max(world_size, 1)}

    # What the per-step exchange is bounded by on one node: every
    # rank contributes its packed observation record; xGMI is point to point
    # (7 links x ~153 GB/s per GPU): with all peers pushing directly a rank
    # receives (N - 1) records over N - 1 links at once, a ring moves
    # (N - 1) / N of the gathered buffer over one link.
    xgmi_bound = None
    if distributed and sharded._packed_global is not None:
        per_rank_bytes = sharded._packed_global.numel() * 4 // max(world_size, 1)
        gathered = per_rank_bytes * world_size
        link = 153e9
        xgmi_bound = {
            "per_rank_bytes": per_rank_bytes, "gathered_bytes": gathered,
            "link_GBps": link / 1e9, "links_per_gpu": 7,
            "direct_ms": per_rank_bytes / link * 1e3 if world_size > 1 else 0.0,
            "ring_ms": (world_size - 1) / max(world_size, 1) * gathered / link * 1e3,
            "note": "lower bounds for the all-gather of the packed observation "
                    "record: direct = every peer pushes its record over its own "
                    "link at once, ring = (N-1)/N of the gathered buffer over one "
                    "link; compare with allgather.ms",
        }

    # ---- per-kernel timing (HIP events on the executor's stream) + roofline ----
    roofline = None
    kernels = []
    if rank == 0:
        stats = sim.profile(args.profile_reps)
        roofline = rooflines(stats, args.sim, args.worlds, ms_per_step)
        if roofline is not None:
            roofline["peak_measured"] = hbm_measured
        kernels = kernel_table(stats, args.sim, args.worlds)

    cpu = None
    if rank == 0 and world_size == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.sim, args.worlds, args.auto_reset_denom, seed,
                           args.settle)

    sharded.close()

    # BASELINE configs[1] (physics off, 4096 worlds) next to the headline
    secondary = None
    if (rank == 0 and world_size == 1 and args.sim == "escape_room_phys"
            and not args.no_secondary):
        secondary = run_single("escape_room", 4096, local_rank, seed,
                               args.auto_reset_denom, 3000, 200, args.profile_reps,
                               settle=args.settle)

    # BASELINE configs[4]: the same worlds + 64x64 RGB-D per agent
    render = None
    if (rank == 0 and world_size == 1 and args.sim == "escape_room_phys"
            and not args.no_secondary):
        render = run_render(args.worlds, local_rank, seed, args.auto_reset_denom,
                            200, 30, 10, min(args.settle, 200),
                            cpu_sample=not args.no_cpu_baseline)

    # What an UNCHANGED simulator gets: the same workloads from the simulators'
    # portable sources (no wave-cooperative extensions; sims/*/sim.cpp
    # SIM_PORTABLE, lib<sim>_portable_hip.so) -- north_star: "existing simulators
    # ... link unchanged".
    portable = None
    if (rank == 0 and world_size == 1 and args.sim == "escape_room_phys"
            and not args.no_secondary
            and os.path.exists(hip_lib_path("escape_room_phys_portable"))):
        portable = {}
        for key, name, w, steps in (("config3", "escape_room_phys_portable",
                                     args.worlds, 400),
                                    ("config2", "escape_room_portable", 4096, 2000)):
            r = run_single(name, w, local_rank, seed, args.auto_reset_denom,
                           steps, 100, 10, settle=args.settle)
            slow = sorted(r["kernels"], key=lambda k: -k["avg_us"])[:4]
            portable[key] = {
                "workload": r["workload"], "value": r["value"], "unit": "steps/s",
                "ms_per_step": r["ms_per_step"],
                "slowest_kernels": [[k["name"], k["avg_us"]] for k in slow],
            }
        portable["note"] = (
            "the bench simulators compiled from their portable sources "
            "(-DSIM_PORTABLE: plain makeEntity / destroyEntity / "
            "findEntitiesWithinAABB, one lane per world); `value` above and "
            "ecs_config2 are the same simulators with the wave-cooperative reset "
            "/ grab systems (sims/*/sim.cpp, SIM_WAVE_API)")

    # BASELINE configs[3]'s one-GPU share (Hide-and-Seek-shaped worlds, 8192 of
    # the 8 x 8192) and configs[0] (the Cartpole plumbing case, 64 worlds: the
    # HIP executor next to the reference's CPU executor it is quoted on)
    hideseek = None
    cartpole = None
    if (rank == 0 and world_size == 1 and args.sim == "escape_room_phys"
            and not args.no_secondary):
        r = run_single("hideseek", 8192, local_rank, seed, args.auto_reset_denom,
                       300, 50, 10, settle=args.settle)
        hideseek = {"workload": r["workload"], "value": r["value"],
                    "unit": "steps/s", "ms_per_step": r["ms_per_step"],
                    "roofline": r["roofline"], "kernels": r["kernels"]}
        cartpole = run_cartpole(local_rank, cpu_sample=not args.no_cpu_baseline)

    dist_world = dist.get_world_size() if distributed else 1
    rccl_ranks = dist_world if distributed and dist.get_backend() == "nccl" else 0
    if distributed:
        dist.barrier()
        dist.destroy_process_group()

    if rank == 0:
        out = {
            "metric": "aggregate env steps/sec",
            "value": value,
            "unit": "steps/s",
            "n_gpus": world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": f"synthetic (worlds advanced {args.settle} steps to their steady "
                    f"state while being set up; {ACTION_WORKLOAD})",
            "config": {
                "workload": workload_fmt.format(w=args.worlds) +
                            f", auto-reset p=1/{args.auto_reset_denom} per world per step",
                "sim": args.sim,
                "worlds_per_gpu": args.worlds,
                "settle_steps": args.settle,
                "total_worlds": total_worlds,
                "dist_world_size": dist_world,
                "rccl_ranks": rccl_ranks,
                "parallelism": f"worlds sharded over {world_size} GPU(s)"
                               + (", one packed RCCL all-gather of the observation "
                                  "tensors per step" if distributed else ""),
            },
            "window_s": round(long_window["seconds"] if long_window else elapsed, 5),
            "timed_steps": long_window["steps"] if long_window else args.steps,
            "short_window": short_window,
            "long_window": long_window,
            "per_rank_ms_per_step": per_rank_ms,
            "allgather": allgather,
            "xgmi_bound": xgmi_bound,
            "hbm_measured": hbm_measured,
            "roofline": roofline,
            "cpu_baseline": cpu,
            "ecs_config2": secondary,
            "render_config5": render,
            "portable_sim": portable,
            "hideseek_config4_share": hideseek,
            "cartpole_config1": cartpole,
            "kernels": kernels,
        }
        # flush what C libraries buffered for "stdout" while it pointed at stderr
        import ctypes
        ctypes.CDLL(None).fflush(None)
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(driver_line(out, write_detail(out))), flush=True)


if __name__ == "__main__":
    main()
