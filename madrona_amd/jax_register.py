"""JAX side of the XLA custom-call bridge.

`register(scope)` is what madrona::py::JAXInterface::setup
(include/madrona/py/bindings.inl) calls with the capsules of a simulator's entry
points, the `sim_encode` opaque (its address) and the interface pytrees; it
registers the custom-call targets with XLA, builds one primitive per entry
point and returns {"init": jitted fn, "step": jitted fn[, "save_ckpts",
"restore_ckpts"]}.

Contract (reference src/python/jax_register.py): operands of every call are
(simulator address constant, ordering token, inputs...), results are
(outputs..., token) -- the token is a zero-element f32 array threaded through
the calls so that XLA neither reorders nor elides them, and it sits FIRST among
the operands and LAST among the results so that the C++ entry point can skip
two buffers and write the rest in order (bindings.hpp: cpuEntryFn `in + 2`,
gpuEntryFn `buffers + 2`).  init() -> {"state", "obs"}; step({"state",
"actions", "resets", "sim_ctrl", "pbt"}) -> {"state", "obs", "rewards",
"dones", "stats", "pbt"}.  The step's result buffers are exactly the tensors
TrainInterface::forEachOutput (include/madrona/py/utils.inl) walks, in its
order: observations..., rewards, dones, stats..., pbt... -- `step_buffer_specs`
below; the reference's script leaves the stats out of its result list while its
C++ side copies them (a simulator that exports a stats tensor would write past
XLA's result array there).

jax is not installed in the build image: this module has only been exercised
up to the import (tests/test_py_bridge.py checks the ImportError); the entry
points themselves are tested by calling the capsules the way XLA does.
"""
from functools import partial

import numpy as np


def step_buffer_specs(outs):
    """(names, specs) of the step call's result buffers from the outputs pytree
    (JAXInterface::outputsToPytree): the order cpuCopyStepOutputs /
    hipCopyStepOutputs write them in.  No jax needed (tests/test_py_bridge.py
    checks it against what the C++ side writes)."""
    obs_names = list(outs["obs"].keys())
    stats_names = list(outs.get("stats", {}).keys())
    pbt_names = list(outs.get("pbt", {}).keys())
    specs = ([outs["obs"][k] for k in obs_names] + [outs["rewards"], outs["dones"]] +
             [outs["stats"][k] for k in stats_names] +
             [outs["pbt"][k] for k in pbt_names])
    return {"obs": obs_names, "stats": stats_names, "pbt": pbt_names}, specs


def step_input_order(ins):
    """Names of the step call's input buffers behind the token: actions...,
    resets, sim_ctrl, pbt... (cpuCopyStepInputs / hipCopyStepInputs)."""
    return (list(ins["actions"].keys()), list(ins.get("pbt", {}).keys()))


def xla_platforms(platform):
    """XLA registry names a custom-call target goes under: jaxlib files 'gpu'
    under CUDA, and a jax-rocm build looks GPU targets up under ROCM."""
    return ["ROCM", "gpu"] if platform == "gpu" else [platform]


def _require_jax():
    try:
        import jax  # noqa: F401
    except ImportError as e:  # pragma: no cover - exercised without jax
        raise ImportError(
            "madrona_amd.jax_register needs jax (jax-rocm for xla_gpu=True); the "
            "simulator's capsules are available without it through "
            "jax(..., register=False)") from e


class _EntryPoint:
    """One XLA custom call + the primitive that lowers to it."""

    def __init__(self, name, capsule, platform, sim_ptr, sim_encode, out_specs):
        import jax
        from jax import core
        from jax.interpreters import mlir, xla
        from jax.lib import xla_client

        self.name = name
        self.sim_ptr = np.uint64(sim_ptr)
        self.sim_encode = sim_encode
        self.out_specs = list(out_specs)     # [(shape, dtype)]
        for registry in xla_platforms(platform):
            try:
                xla_client.register_custom_call_target(name, capsule, platform=registry)
            except Exception:   # (a jaxlib without that registry)
                if registry == xla_platforms(platform)[-1]:
                    raise

        prim = core.Primitive(name)
        prim.multiple_results = True
        prim.def_impl(partial(xla.apply_primitive, prim))
        prim.def_abstract_eval(self._abstract)
        mlir.register_lowering(prim, self._lower, platform=platform)
        self.primitive = prim
        self._jax = jax

    def _abstract(self, *_):
        from jax.core import ShapedArray
        token = ShapedArray((0,), np.float32)
        return (token, *(ShapedArray(tuple(s), np.dtype(d))
                         for s, d in self.out_specs))

    def _lower(self, ctx, *operands):
        from jax.interpreters import mlir
        from jax.interpreters.mlir import ir, dtype_to_ir_type
        from jaxlib.hlo_helpers import custom_call

        def rowmajor(rank):
            return tuple(range(rank - 1, -1, -1))

        token_type = ir.RankedTensorType.get((0,), dtype_to_ir_type(np.dtype("float32")))
        if operands:
            token, *inputs = operands
        else:   # init: no state yet
            token, inputs = mlir.ir_constant(np.empty((0,), np.float32)), []
        in_layouts = [rowmajor(ir.RankedTensorType(i.type).rank) for i in inputs]
        out_types = [ir.RankedTensorType.get(tuple(s), dtype_to_ir_type(np.dtype(d)))
                     for s, d in self.out_specs]
        out_layouts = [rowmajor(len(s)) for s, _ in self.out_specs]
        results = custom_call(
            self.name,
            backend_config=self.sim_encode,
            operands=[mlir.ir_constant(self.sim_ptr), token, *inputs],
            operand_layouts=[(), (0,), *in_layouts],
            result_types=[*out_types, token_type],
            result_layouts=[*out_layouts, (0,)],
            has_side_effect=True,
        ).results
        *outs, token = results
        return (token, *outs)

    def bind(self, *flat_inputs):
        state, *outs = self.primitive.bind(*flat_inputs)
        return state, outs


def register(scope):
    _require_jax()
    import jax

    platform = scope["platform"]
    prefix = f"{type(scope['sim_obj']).__name__}_{id(scope['sim_obj'])}"
    ins, outs = scope["step_inputs_iface"], scope["step_outputs_iface"]
    out_names, step_specs = step_buffer_specs(outs)
    obs_names, stats_names, pbt_out_names = (out_names["obs"], out_names["stats"],
                                             out_names["pbt"])
    obs_specs = [outs["obs"][k] for k in obs_names]
    action_names, pbt_in_names = step_input_order(ins)

    def entry(kind, specs):
        return _EntryPoint(f"{prefix}_{kind}", scope[kind], platform,
                           scope["sim_ptr"], scope["sim_encode"], specs)

    init_entry = entry("init", obs_specs)
    step_entry = entry("step", step_specs)

    def init_func():
        state, flat = init_entry.bind()
        return {"state": state, "obs": dict(zip(obs_names, flat))}

    def step_func(step_inputs):
        flat_in = [step_inputs["state"]]
        flat_in += [step_inputs["actions"][k] for k in action_names]
        flat_in += [step_inputs["resets"], step_inputs["sim_ctrl"]]
        flat_in += [step_inputs["pbt"][k] for k in pbt_in_names]
        state, flat = step_entry.bind(*flat_in)
        n, m = len(obs_names), len(stats_names)
        return {"state": state, "obs": dict(zip(obs_names, flat[:n])),
                "rewards": flat[n], "dones": flat[n + 1],
                "stats": dict(zip(stats_names, flat[n + 2:n + 2 + m])),
                "pbt": dict(zip(pbt_out_names, flat[n + 2 + m:]))}

    fns = {"init": jax.jit(init_func), "step": jax.jit(step_func)}

    ckpt = scope.get("ckpt_iface")
    if ckpt is not None and "save_ckpts" in scope:
        save_entry = entry("save_ckpts", [ckpt["data"]])
        restore_entry = entry("restore_ckpts", obs_specs)

        def save_ckpts_func(save_inputs):
            state, flat = save_entry.bind(save_inputs["state"],
                                          save_inputs["should_save"])
            return {"state": state, "ckpts": flat[0]}

        def restore_ckpts_func(restore_inputs):
            state, flat = restore_entry.bind(restore_inputs["state"],
                                             restore_inputs["should_restore"],
                                             restore_inputs["ckpt_data"])
            return {"state": state, "obs": dict(zip(obs_names, flat))}

        fns["save_ckpts"] = jax.jit(save_ckpts_func)
        fns["restore_ckpts"] = jax.jit(restore_ckpts_func)
    return fns
