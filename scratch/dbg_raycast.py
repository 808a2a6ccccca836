import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from madrona_amd.simlib import Simulator, hip_lib_path
from raycast_utils import *
from test_raycast_gpu import _sim_geometry, _offsets
worlds, res = 37, 32
with Simulator(hip_lib_path("render_prep"), worlds, seed=11, flags=1 | (res << 8)) as hip:
    geo = _sim_geometry(hip)
    hip.step(1)
    hip.render()
    d = hip.dump_all()
    inst, ic = d["Renderable.InstanceData"]
    views, vc = d["Camera.PerspectiveCameraData"]
    lights, lc = d["Light.LightDesc"]
    rr, rd = ref_render(geo, worlds, inst, _offsets(ic), ic, views, lights, _offsets(lc), lc, res, threads=16)
    hd = hip.read_tensor("depth")
    os.makedirs("gpurun_out", exist_ok=True)
    np.savez("gpurun_out/dbg_raycast.npz", inst=inst, ic=ic, views=views, lights=lights, lc=lc, hd=hd, hrgb=hip.read_tensor("rgb"), tlb=d["Renderable.TLBVHNode"][0], mort=d["Renderable.MortonCode"][0])
    for v in range(2 * worlds):
        hit_h, hit_r = hd[v] > 0, rd[v] > 0
        fl = int((hit_h != hit_r).sum())
        both = hit_h & hit_r
        rel = np.abs(hd[v][both] - rd[v][both]) / rd[v][both]
        far = int((rel > 1e-5).sum())
        if fl or far:
            print("view", v, "world", v // 2, "n_inst", ic[v // 2], "flipped", fl, "hip-only", int((hit_h & ~hit_r).sum()),
                  "ref-only", int((~hit_h & hit_r).sum()), "far", far, "hip<ref", int((hd[v][both] < rd[v][both] * (1 - 1e-5)).sum()))
    print("counts", ic.tolist())
    mort = d["Renderable.MortonCode"][0].view(np.uint32).ravel()
    off = _offsets(ic)
    w = [v // 2 for v in range(2 * worlds) if ((hd[v] > 0) != (rd[v] > 0)).any()]
    if w:
        w = w[0]
        print("world", w, "morton", [hex(x) for x in mort[off[w]:off[w] + ic[w]]])
        ii = inst.view(INSTANCE_DT).ravel()[off[w]:off[w] + ic[w]]
        print(ii["objectID"], ii["position"], ii["scale"])
