"""flowmap_amd.install(graph=True): ModelWrapperOverfit.training_step replayed as hipGraphs (flowmap_amd/training.py).

CPU: the rebinding, the phase signature and the state machine around the graphs (eager warm-up steps in a new phase, capture, replay,
drop on a phase change, a failed capture keeps the eager step for good) with the capture itself replaced by a recording double — on the
host double nothing is ever captured and the installed step is the package's own.  GPU: the real thing against the eager installed
step, through a trainer's order of calls (training_step → zero_grad → backward → optimiser step), across a loss switching on and a
phase change."""

import pytest
import torch

from test_install_standin import _problem


def _wrapper(name, with_tracks, dev, enable_tracking_after=0, lr=1e-3, softmin=None):
    from flowmap.model.model_wrapper_overfit import ModelWrapperOverfit, ModelWrapperOverfitCfg

    g, model, batch, flows, tracks, losses = _problem(name, with_tracks, dev)
    if softmin is not None:  # the reference's default intrinsics (config/overfit.yaml): the softmin sweep, handing over to a regressed focal length
        from flowmap.model.intrinsics import IntrinsicsSoftminCfg, RegressionCfg, get_intrinsics

        after_step, window = softmin
        torch.manual_seed(5)
        model.intrinsics = get_intrinsics(IntrinsicsSoftminCfg("softmin", 200, 0.5, 2.0, 20, RegressionCfg(after_step, window))).to(dev)
    if with_tracks:
        losses[1].cfg.enable_after = enable_tracking_after
    wrapper = ModelWrapperOverfit(ModelWrapperOverfitCfg(lr, 32), model, batch, flows, tracks, losses, [])
    wrapper.train()
    return wrapper


def test_install_graph_rebinds_training_step_and_uninstall_restores_it(standin):
    import flowmap.model.model_wrapper_overfit as ref_wrapper

    import flowmap_amd
    from flowmap_amd import _ops

    original = ref_wrapper.ModelWrapperOverfit.training_step
    flowmap_amd.install()
    assert ref_wrapper.ModelWrapperOverfit.training_step is original  # off by default
    flowmap_amd.install(graph=True)
    try:
        assert ref_wrapper.ModelWrapperOverfit.training_step is not original
        assert ref_wrapper.ModelWrapperOverfit.training_step.__wrapped__ is original
        assert _ops.options.tap_image is True  # (untouched: graphs exist only below the size at which the tap exchange engages)
    finally:
        flowmap_amd.uninstall()
    assert ref_wrapper.ModelWrapperOverfit.training_step is original


def test_on_the_host_double_the_installed_graph_step_is_the_packages_own(standin):
    """No GPU, no graphs: install(graph=True) must cost nothing and change nothing there."""
    import flowmap_amd
    from flowmap_amd import _lib
    from helpers import build_host_sim

    _lib.set_library_for_testing(build_host_sim())
    try:
        histories = {}
        for graph in (False, True):
            flowmap_amd.install(graph=graph)
            wrapper = _wrapper("step_scene_flow_tracking", True, "cpu", enable_tracking_after=1)
            optimizer = wrapper.configure_optimizers()
            history = []
            for _ in range(4):
                history.append(float(wrapper.fit_steps(optimizer, 1).detach()))
            histories[graph] = history
            state = wrapper.__dict__.get("_fm_graphed_training")
            assert (state is not None) == graph
            if graph:
                assert state.captures == 0 and state.replays == 0 and state.disabled is None
            assert set(wrapper.logged) == {"train/loss/flow", "train/loss/tracking"}
            flowmap_amd.uninstall()
        assert histories[True] == histories[False]
        assert histories[False][-1] < histories[False][1]
    finally:
        flowmap_amd.uninstall()
        _lib.set_library_for_testing(None)


class _RecordingGraph:
    def __init__(self, run):
        self.run, self.replays = run, 0

    def replay(self):
        self.replays += 1
        self.run()


def _recording_training(eager):
    """GraphedTraining with the hipGraph capture replaced by closures that re-run the step's two halves (the control flow is what is tested)."""
    from flowmap_amd import training

    class Recording(training.GraphedTraining):
        fail_capture = False

        def on_device(self, wrapper):
            return torch.is_grad_enabled() and wrapper.training

        def capture(self, wrapper):
            if self.fail_capture:
                raise RuntimeError("no capture today")
            self.params = [p for p in wrapper.parameters() if p.requires_grad]
            holder = {}

            def forward():
                total, values, _ = self.forward(wrapper)
                holder["total"] = total
                for dst, src in zip(self.values, values):
                    dst.copy_(src.detach())
                self.total.data.copy_(total.detach())

            def backward():
                grads = torch.autograd.grad(holder["total"], self.params, allow_unused=True)
                for dst, src in zip(self.grads, grads):
                    if dst is not None:
                        dst.copy_(src)

            total, values, self.errors = self.forward(wrapper)
            self.values = [v.detach().clone() for v in values]
            self.grads = [None if g is None else g.clone() for g in torch.autograd.grad(total, self.params, allow_unused=True)]
            made = total.detach().clone().requires_grad_(True)
            made.__class__ = training.GraphedLoss
            made.__dict__["_fm_graphed_training"] = self
            self.total = made
            self.forward_graph, self.backward_graph = _RecordingGraph(forward), _RecordingGraph(backward)
            self.captures += 1

    return Recording(eager)


def test_the_state_machine_around_the_graphs(standin):
    import flowmap.model.model_wrapper_overfit as ref_wrapper

    import flowmap_amd
    from flowmap_amd import _lib, training
    from helpers import build_host_sim

    _lib.set_library_for_testing(build_host_sim())
    try:
        flowmap_amd.install()
        eager = ref_wrapper.ModelWrapperOverfit.training_step
        reference = _wrapper("step_scene_flow_tracking", True, "cpu", enable_tracking_after=2)
        ref_opt = torch.optim.Adam(reference.parameters(), lr=1e-3)
        wrapper = _wrapper("step_scene_flow_tracking", True, "cpu", enable_tracking_after=2)
        opt = torch.optim.Adam(wrapper.parameters(), lr=1e-3)
        state = _recording_training(eager)
        trail = []

        def one_step():
            loss = state(wrapper, None)
            trail.append(type(loss).__name__)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            wrapper.global_step += 1
            return float(loss.detach())

        ours = [one_step() for _ in range(8)]
        theirs = [float(reference.fit_steps(ref_opt, 1).detach()) for _ in range(8)]
        # steps 0-1: the tracking loss is still off (no phase); 2-3: the new phase's eager steps; 4: capture + first replay; 5-7: replays
        assert trail[:4] != ["GraphedLoss"] * 4 and "GraphedLoss" not in trail[:4] and trail[4:] == ["GraphedLoss"] * 4, trail
        assert state.captures == 1 and state.replays == 4
        assert ours == pytest.approx(theirs, rel=1e-6)
        for a, b in zip(wrapper.parameters(), reference.parameters()):
            assert torch.allclose(a, b, rtol=1e-6, atol=1e-9)
        assert set(wrapper.logged) == {"train/loss/flow", "train/loss/tracking"}
        assert float(wrapper.logged["train/loss/flow"]) + float(wrapper.logged["train/loss/tracking"]) == pytest.approx(ours[-1], rel=1e-6)

        # the loss of a replayed step: trainer arithmetic that changes nothing passes, anything else is refused loudly
        loss = state(wrapper, None)
        assert (loss / 1) is loss and (1 * loss) is loss and (0 + loss) is loss
        for bad in (lambda: loss / 2, lambda: loss * 0.5, lambda: loss + 1.0, lambda: -loss, lambda: loss.backward(gradient=torch.ones(()))):
            with pytest.raises(RuntimeError, match="install\\(graph=True\\)"):
                bad()
        loss.backward()
        with pytest.raises(RuntimeError, match="twice"):
            loss.backward()
        wrapper.global_step += 1

        # a phase change: other tracks objects -> the graphs go, two eager steps, a new capture
        wrapper.tracks = list(wrapper.tracks)
        trail.clear()
        for _ in range(4):
            one_step()
        assert trail == ["RootLoss", "RootLoss", "GraphedLoss", "GraphedLoss"] or ("GraphedLoss" not in trail[:2] and trail[2:] == ["GraphedLoss"] * 2), trail
        assert state.captures == 2
        with pytest.raises(RuntimeError, match="dropped"):
            loss.backward()  # the loss of the old phase's graphs

        # eval() / no_grad: the package's own step, graphs dropped
        wrapper.eval()
        assert type(state(wrapper, None)).__name__ != "GraphedLoss" and state.forward_graph is None
        wrapper.train()

        # a capture that fails: one warning, the eager step for good
        failing = _recording_training(eager)
        failing.fail_capture = True
        with pytest.warns(UserWarning, match="capturing training_step failed"):
            for _ in range(3):
                out = failing(wrapper, None)
        assert failing.disabled is not None and type(out).__name__ != "GraphedLoss"
        assert type(failing(wrapper, None)).__name__ != "GraphedLoss" and failing.captures == 0

        # in-pass Adam (FusedAdam.fuse_depth_update) and a softmin module before its hand-over have no constant phase
        assert state.signature(wrapper) is not None
        wrapper.model.backbone.depth.__dict__["_fm_fused_adam"] = object()
        assert state.signature(wrapper) is None
        del wrapper.model.backbone.depth.__dict__["_fm_fused_adam"]
        wrapper.losses[1].cfg.enable_after = wrapper.global_step + 1
        assert state.signature(wrapper) is None
        wrapper.losses[1].cfg.enable_after = 0
        handle = wrapper.model.backbone.depth.register_hook(lambda g: g)  # a hook on a parameter's gradient: a replay would not run it
        assert state.signature(wrapper) is None
        handle.remove()
        assert state.signature(wrapper) is not None
        from flowmap_amd import _ops

        previous, _ops.options.tap_exchange_min_bytes = _ops.options.tap_exchange_min_bytes, wrapper.model.backbone.depth.numel() * 4
        try:  # depth maps from the size at which the tap exchange engages: HBM-bound, the eager step stays
            assert state.signature(wrapper) is None
        finally:
            _ops.options.tap_exchange_min_bytes = previous
        assert state.signature(wrapper) is not None
        assert isinstance(training.make_training_step(eager).__wrapped__, type(eager))
    finally:
        flowmap_amd.uninstall()
        _lib.set_library_for_testing(None)


def _run_trainer(graph, dev, steps_a=8, steps_b=5):
    import flowmap_amd

    flowmap_amd.install(graph=graph)
    try:
        wrapper = _wrapper("step_scene_flow_tracking", True, dev, enable_tracking_after=2)
        optimizer = wrapper.configure_optimizers()
        assert type(optimizer).__name__ == "FusedAdam"
        history = [float(wrapper.fit_steps(optimizer, 1).detach()) for _ in range(steps_a)]
        logged = {k: float(v) for k, v in wrapper.logged.items()}
        wrapper.tracks = list(wrapper.tracks)  # a phase change: the graphs are dropped and captured again
        history += [float(wrapper.fit_steps(optimizer, 1).detach()) for _ in range(steps_b)]
        state = wrapper.__dict__.get("_fm_graphed_training")
        return history, logged, [p.detach().clone() for p in wrapper.parameters()], state
    finally:
        flowmap_amd.uninstall()


@pytest.mark.gpu
def test_the_replayed_training_step_follows_the_eager_installed_one(standin):
    """13 trainer iterations with FusedAdam: the tracking loss switches on at step 2 (no phase before), steps 2-3 run eagerly, step 4 is
    captured, 4-7 replayed; new tracks objects at step 8: two eager steps, a second capture, replays.  Loss history, logged values and every
    parameter follow the eager installed run."""
    from conftest import assert_close

    eager_history, eager_logged, eager_params, no_state = _run_trainer(False, "cuda:0")
    history, logged, params, state = _run_trainer(True, "cuda:0")
    assert no_state is None and state is not None and state.disabled is None, getattr(state, "disabled", None)
    assert state.captures == 2 and state.replays == (8 - 4) + (5 - 2)
    assert_close(torch.tensor(history), torch.tensor(eager_history), 1e-5, what="loss history")
    assert eager_history[-1] < eager_history[2]  # it optimises
    assert set(logged) == set(eager_logged) == {"train/loss/flow", "train/loss/tracking"}
    for key in logged:
        assert abs(logged[key] - eager_logged[key]) <= 1e-5 * abs(eager_logged[key]), key
    for ours, theirs in zip(params, eager_params):
        assert_close(ours, theirs, 1e-5, what="parameters")
    from flowmap_amd import _ops

    assert _ops.graph_capturable is False  # on for the capture only: a process-wide switch must not outlive it (later eager sweeps draw from torch's generator)


@pytest.mark.gpu
def test_a_replayed_training_step_enqueues_two_graph_launches(standin):
    """After the capture a step's forward + losses and its backward are one hipGraph launch each: no kernel is enqueued one by one."""
    import flowmap_amd

    flowmap_amd.install(graph=True)
    try:
        wrapper = _wrapper("step_scene_flow_tracking", True, "cuda:0")
        wrapper.fit_steps(None, 3)
        state = wrapper.__dict__["_fm_graphed_training"]
        assert state.captures == 1 and state.disabled is None
        from torch.utils._python_dispatch import TorchDispatchMode

        seen = []

        class Watch(TorchDispatchMode):
            def __torch_dispatch__(self, func, types, args=(), kwargs=None):
                seen.append(str(func))
                return func(*args, **(kwargs or {}))

        with Watch():
            wrapper.fit_steps(None, 1)
        assert not [name for name in seen if "flowmap_amd" in name], seen  # no operator of the library was dispatched: the graphs carry them
        assert wrapper.model.backbone.depth.grad is not None and torch.isfinite(wrapper.model.backbone.depth.grad).all()
    finally:
        flowmap_amd.uninstall()


def test_the_step_on_parameter_aliases_is_the_step(standin):
    """What the capture differentiates: the model called through torch.func.functional_call on ALIASES of its parameters (new leaves over the
    same storage that carry the parameter's derived constants).  On the host double: same loss, same gradients as the eager installed step,
    the weights' gradient in the parameter's own arena, the plans of the eager steps reused (no new scatter plan is built)."""
    import flowmap.model.model_wrapper_overfit as ref_wrapper

    import flowmap_amd
    from flowmap_amd import _lib, _ops, training
    from helpers import build_host_sim

    _lib.set_library_for_testing(build_host_sim())
    try:
        flowmap_amd.install()
        wrapper = _wrapper("step_scene_flow_tracking", True, "cpu")
        wrapper.fit_steps(None, 2)  # the eager steps of a phase: packed inputs, scatter / tap plans
        wrapper.zero_grad(set_to_none=True)
        loss = wrapper.training_step(None)
        loss.backward()
        expected = {name: p.grad.clone() for name, p in wrapper.model.named_parameters()}
        wrapper.zero_grad(set_to_none=True)
        before = dict(_ops.counters)
        state = training.GraphedTraining(ref_wrapper.ModelWrapperOverfit.training_step)
        named = list(wrapper.model.named_parameters())
        aliases = state.alias_parameters(named)
        total, values, _ = state.forward_on_aliases(wrapper, aliases)
        grads = torch.autograd.grad([total], [aliases[name] for name, _ in named], [_ops.unit_seed(total.device)], allow_unused=True)
        assert float(total.detach()) == pytest.approx(float(loss.detach()), rel=1e-6) and len(values) == 2
        for (name, p), g in zip(named, grads):
            assert p.grad is None and g is not None, name
            assert torch.allclose(g, expected[name], rtol=1e-5, atol=1e-12), name
        moved = {k for k, v in _ops.counters.items() if v != before.get(k, 0)}
        assert not {k for k in moved if "plan" in k and "hit" not in k}, moved
    finally:
        flowmap_amd.uninstall()
        _lib.set_library_for_testing(None)


def _run_softmin_trainer(graph, dev, steps=10):
    import flowmap_amd

    flowmap_amd.install(graph=graph)
    try:
        torch.manual_seed(11)
        wrapper = _wrapper("step_scene_flow_tracking", True, dev, softmin=(3, 2))
        optimizer = wrapper.configure_optimizers()
        history = [float(wrapper.fit_steps(optimizer, 1).detach()) for _ in range(steps)]
        return history, [p.detach().clone() for p in wrapper.parameters()], wrapper.__dict__.get("_fm_graphed_training")
    finally:
        flowmap_amd.uninstall()


@pytest.mark.gpu
def test_the_replay_starts_after_the_softmin_hand_over(standin):
    """The reference's default intrinsics: the softmin sweep (steps 0-2, random pixels, the focal lengths of the window recorded on the host), the
    hand-over to the regressed focal length at step 3 — all of that runs as the package's own step; steps 4-5 are the new phase's eager steps,
    step 6 is captured, 6-9 replayed.  Same trajectory as the eager installed run."""
    from conftest import assert_close

    eager_history, eager_params, _ = _run_softmin_trainer(False, "cuda:0")
    history, params, state = _run_softmin_trainer(True, "cuda:0")
    assert state is not None and state.disabled is None, getattr(state, "disabled", None)
    assert state.captures == 1 and state.replays == 4
    assert_close(torch.tensor(history), torch.tensor(eager_history), 1e-5, what="loss history")
    for ours, theirs in zip(params, eager_params):
        assert_close(ours, theirs, 1e-5, what="parameters")


@pytest.mark.gpu
def test_the_losses_as_branches_of_the_captured_graph(standin, monkeypatch):
    """FLOWMAP_AMD_GRAPH_STREAMS=1 (off by default: 2 % at 180x240): the tracking loss captured on a stream of its own beside the flow loss, its
    backward on that stream — the same trajectory as the eager installed run."""
    from conftest import assert_close

    eager_history, _, eager_params, _ = _run_trainer(False, "cuda:0", steps_a=7, steps_b=0)
    monkeypatch.setenv("FLOWMAP_AMD_GRAPH_STREAMS", "1")
    history, _, params, state = _run_trainer(True, "cuda:0", steps_a=7, steps_b=0)
    assert state.concurrent_losses and state.disabled is None and state.captures == 1 and state.replays == 3, state.disabled
    assert_close(torch.tensor(history), torch.tensor(eager_history), 1e-5, what="loss history")
    for ours, theirs in zip(params, eager_params):
        assert_close(ours, theirs, 1e-5, what="parameters")
