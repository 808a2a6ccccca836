"""What physicsOrderKernel believed of every world against what the step took
(needs a build with -DMADRONA_PHYS_WAVELOG=1:

    make -C madrona_amd OUT=_build_wavelog EXTRA=-DMADRONA_PHYS_WAVELOG=1 runtime \\
         _build_wavelog/libescape_room_phys_hip.so

the kernel then writes eight words per world to ecs_state::moduleData[1]).
phys_wave_log.py [SIM] [WORLDS] [SLOTS]"""
import sys, os, ctypes as C, heapq
os.environ.setdefault('MADRONA_HIP_BUILD_DIR', '_build_wavelog')
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from madrona_amd.simlib import Simulator, hip_lib_path, runtime_lib
SIM = sys.argv[1] if len(sys.argv) > 1 else 'escape_room_phys'
W = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
SLOTS = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
A = 2 if SIM.startswith('escape') else 5


def list_schedule(times, slots):
    """Greedy dispatch in the given order onto `slots` slots: makespan."""
    free = [0.0] * slots
    heapq.heapify(free)
    end = 0.0
    for t in times:
        s = heapq.heappop(free)
        heapq.heappush(free, s + t)
        end = max(end, s + t)
    return end


with Simulator(hip_lib_path(SIM), W, seed=5, flags=200) as hip:
    rt = runtime_lib()
    rt.mwhip_alloc_device.restype = C.c_void_p
    rt.mwhip_alloc_device.argtypes = [C.c_void_p, C.c_uint64, C.c_int]
    rt.mwhip_set_module_data.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    rt.mwhip_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    import torch
    rng = np.random.default_rng(0)
    if SIM.startswith('escape'):
        ring = np.stack([np.stack([rng.integers(0, 4, (W, A)), rng.integers(0, 8, (W, A)),
                                   rng.integers(-2, 3, (W, A)), rng.integers(0, 2, (W, A))], -1)
                         for _ in range(61)]).astype(np.int32)
        dev = torch.from_numpy(ring).cuda()
        hip.set_input_ring('action', dev.data_ptr(), 61)
    hip.step(300)
    buf = rt.mwhip_alloc_device(hip.hip_exec(), W * 32, 1)
    rt.mwhip_set_module_data(hip.hip_exec(), 1, buf)
    prev = None
    for it in range(6):
        hip.step(1)
        rec = np.zeros((W, 8), np.uint32)
        rt.mwhip_memcpy_d2h(rec.ctypes.data, buf, W * 32)
        job, pred, work, wtime, t0, cb, hc, cost = (rec[:, i].astype(np.int64) for i in range(8))
        bodies, cands = cb >> 16, cb & 0xFFFF
        hulls, contacts = hc & 0xFFFF, hc >> 16
        it_max, it_shared = work & 0xFFFF, work >> 16
        print(f"  hull-hull rounds per wavefront and step: each half its own pairs {it_max.mean():.2f} "
              f"(heaviest tenth of the wavefronts {np.sort(it_max)[-len(it_max) // 10:].mean():.2f}), "
              f"halves sharing them out {it_shared.mean():.2f} ({np.sort(it_shared)[-len(it_shared) // 10:].mean():.2f}); "
              f"corr(wave time, rounds) {np.corrcoef(it_max, wtime)[0, 1]:.3f}")
        # per wave (job): the time, the start (100 MHz ticks -> us)
        njobs = int(job.max()) + 1
        jt = np.zeros(njobs); js = np.zeros(njobs); jp = np.zeros(njobs)
        jt[job] = wtime / 100.0
        js[job] = ((t0 - t0.min()) & 0xFFFFFFFF) / 100.0
        np.maximum.at(jp, job, pred / 100.0)
        kernel = (js + jt).max()
        print(f"step {it}: {njobs} jobs, kernel {kernel:.1f} us (first start to last end), "
              f"sum of wave times {jt.sum() / 1e3:.1f} ms = {jt.sum() / kernel:.0f} in flight; "
              f"wave time min/mean/p90/p99/max {jt.min():.0f}/{jt.mean():.0f}/"
              f"{np.percentile(jt, 90):.0f}/{np.percentile(jt, 99):.0f}/{jt.max():.0f}")
        print(f"  corr(predicted, this step's cost) per world {np.corrcoef(pred, cost)[0, 1]:.3f}; "
              f"per wave: corr(max predicted of the pair, wave time) {np.corrcoef(jp, jt)[0, 1]:.3f}")
        for name, v in (("hull pairs", hulls), ("contacts", contacts), ("candidates", cands), ("bodies", bodies)):
            print(f"  corr(this step's cost, {name}) {np.corrcoef(cost, v)[0, 1]:.3f}", end=";")
        print()
        # list scheduling with the measured wave times (no contention model)
        order = np.arange(njobs)
        print(f"  greedy dispatch of the measured times on {SLOTS} slots: as dispatched "
              f"{list_schedule(jt[order], SLOTS):.1f} us, longest first "
              f"{list_schedule(np.sort(jt)[::-1], SLOTS):.1f}, shortest first "
              f"{list_schedule(np.sort(jt), SLOTS):.1f}, random "
              f"{list_schedule(rng.permutation(jt), SLOTS):.1f}; lower bound "
              f"{max(jt.sum() / SLOTS, jt.max()):.1f}")
        # who finishes last
        last = np.argsort(js + jt)[-10:]
        print("  last ten to finish (job, start, time, predicted): " +
              "; ".join(f"{j} {js[j]:.0f} {jt[j]:.0f} {jp[j]:.0f}" for j in last))
        parts = np.array_split(jt, 16)
        print("  wave time by sixteenth of the order: " + " ".join(f"{p.mean():.0f}" for p in parts))
        parts = np.array_split(jp, 16)
        print("  predicted by sixteenth of the order: " + " ".join(f"{p.mean():.0f}" for p in parts))
        if prev is not None:
            print(f"  corr(this step's cost, last step's cost) per world {np.corrcoef(cost, prev[0])[0, 1]:.3f}; "
                  f"hull pairs {np.corrcoef(hulls, prev[1])[0, 1]:.3f}; contacts {np.corrcoef(contacts, prev[2])[0, 1]:.3f}")
            # a linear model of the wave time from the pair's counts
        X = np.stack([np.ones(W), hulls, contacts, cands, bodies], 1).astype(np.float64)
        # per wave: max over the pair of each count
        idx = np.argsort(job, kind='stable')
        if W % 2 == 0:
            Xw = np.maximum(X[idx][0::2], X[idx][1::2])
            coef, res, *_ = np.linalg.lstsq(Xw, jt, rcond=None)
            fit = Xw @ coef
            print(f"  wave time ~ {coef[0]:.0f} + {coef[1]:.1f} hull pairs + {coef[2]:.2f} contacts + "
                  f"{coef[3]:.2f} candidates + {coef[4]:.1f} bodies (max of the pair): corr {np.corrcoef(fit, jt)[0, 1]:.3f}")
        prev = (cost, hulls, contacts)
