"""CPU-only: the C-ABI library loads and exports every symbol include/mwhip.h
declares; simulator libraries export the simulator C API.  No compute calls."""
import ctypes as C
import os
import re

import pytest

from madrona_amd.simlib import HIP_BUILD_DIR, REPO_ROOT, hip_lib_path

SIMS = ["cartpole", "escape_room", "sort_stress", "escape_room_phys", "hideseek",
        "ball_pit", "broadphase_only", "render_prep", "escape_room_render"]


def declared_functions(header_path):
    text = open(header_path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mwhip_[a-z0-9_]+)\s*\(", text)))


def test_runtime_exports_every_declared_symbol(built):
    lib = C.CDLL(os.path.join(HIP_BUILD_DIR, "libmadrona_hip.so"), mode=C.RTLD_GLOBAL)
    names = declared_functions(os.path.join(REPO_ROOT, "include", "mwhip.h"))
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in mwhip.h but not exported: {missing}"


def test_header_is_plain_c(built, tmp_path):
    """include/mwhip.h must compile as C (plain pointers and sizes only)."""
    import subprocess
    src = tmp_path / "abi_check.c"
    src.write_text('#include "mwhip.h"\nint main(void) { return (int)sizeof(mwhip_node_desc) == 0; }\n')
    res = subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I",
                          os.path.join(REPO_ROOT, "include"), str(src), "-c", "-o",
                          str(tmp_path / "abi_check.o")], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr


@pytest.mark.parametrize("sim", SIMS)
def test_simulator_library_exports_c_api(built, sim):
    C.CDLL(os.path.join(HIP_BUILD_DIR, "libmadrona_hip.so"), mode=C.RTLD_GLOBAL)
    lib = C.CDLL(hip_lib_path(sim))
    for fn in ["sim_create", "sim_destroy", "sim_step", "sim_tensor_ptr",
               "sim_column_dump", "sim_column_dump_raw", "sim_hip_run_taskgraph",
               "sim_hip_exec", "sim_hip_step_graph", "madronaMWHipUserEntry"]:
        assert hasattr(lib, fn), fn


def test_error_reporting_without_creating_an_executor(built):
    lib = C.CDLL(os.path.join(HIP_BUILD_DIR, "libmadrona_hip.so"), mode=C.RTLD_GLOBAL)
    lib.mwhip_last_error.restype = C.c_char_p
    assert isinstance(lib.mwhip_last_error(), bytes)


def test_physics_test_shims_are_built(built):
    """The function-level physics test libraries (host flavour for the CPU
    tests, device flavour for -m gpu) exist and export their entry points."""
    C.CDLL(os.path.join(HIP_BUILD_DIR, "libmadrona_hip.so"), mode=C.RTLD_GLOBAL)
    host = C.CDLL(os.path.join(HIP_BUILD_DIR, "libphys_host_test.so"))
    assert hasattr(host, "amd_bake_objects") and hasattr(host, "amd_collide_pair")
    dev = C.CDLL(os.path.join(HIP_BUILD_DIR, "libphys_device_test.so"))
    assert hasattr(dev, "dev_collide_pairs") and hasattr(dev, "dev_reference_kats")


def test_overlay_headers_of_the_boundary_exist(built):
    """Every header SURVEY 8b says simulators / Managers / bindings include is
    part of the overlay (their contents are compiled and exercised by
    tests/test_api_conformance.py)."""
    inc = os.path.join(REPO_ROOT, "madrona_amd", "include", "madrona")
    for rel in ["sync.hpp", "memory.hpp", "dyn_array.hpp", "heap_array.hpp",
                "inline_array.hpp", "mw_gpu/const.hpp", "mw_gpu/host_print.hpp",
                "py/utils.hpp", "mw_gpu.hpp", "mw_gpu_entry.hpp", "taskgraph_builder.hpp",
                "custom_context.hpp", "physics.hpp", "physics_loader.hpp"]:
        assert os.path.exists(os.path.join(inc, rel)), rel
    conf = C.CDLL(os.path.join(HIP_BUILD_DIR, "libapi_conformance.so"))
    for fn in ["conf_containers", "conf_tensor_bytes", "conf_run",
               "madronaMWHipUserEntry"]:
        assert hasattr(conf, fn), fn


def test_node_data_limit_is_one_number(built):
    """A node's data block: the limit the builder asserts at compile time
    (TaskGraph::maxNodeDataBytes) is the one the C ABI enforces
    (MWHIP_MAX_NODE_DATA_BYTES), and the physics step's node -- its parameters
    with the per-launch frame of addresses behind them -- fits."""
    header = open(os.path.join(REPO_ROOT, "include", "mwhip.h")).read()
    abi = int(re.search(r"#define\s+MWHIP_MAX_NODE_DATA_BYTES\s+(\d+)u", header).group(1))
    tg = open(os.path.join(REPO_ROOT, "madrona_amd", "include", "madrona",
                           "taskgraph.hpp")).read()
    builder = int(re.search(r"maxNodeDataBytes\s*=\s*(\d+);", tg).group(1))
    assert abi == builder == 2048
    # sizeof(PhysicsStepNode) from its members (8 rigid-body archetypes at most)
    max_arch = 8
    frame = 4 + 4 + 4 * max_arch + 2 * 8 * max_arch + 8 * max_arch * 14 + 4 * 8 + 7 * 8 + 4 * 8
    params = 4 + 4 + 5 * 8 + 4 + 4
    assert frame + params <= abi
