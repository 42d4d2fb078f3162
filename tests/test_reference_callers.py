"""The reference's own CALLER code - `fiery/trainer.py` (TrainingModule.shared_step, prepare_future_labels), `fiery/losses.py`,
`fiery/metrics.py`, `fiery/utils/instance.py` and the body of `evaluate.py` - run UNMODIFIED against `fiery_amd.model.Fiery`
through the one-line import swap of INTEGRATION.md, beside the same code on the reference's own class ("drops
into train.py / evaluate.py unchanged").  Build container only (`needs_reference`): the product class runs on the CPU
simulator of its kernel sources; third-party packages the reference imports are stand-ins (oracle/ref_shims.py)."""
import os

import pytest
import torch

from tests import parity_report
from tests.helpers import randomise_weights, tiny_cfg

pytestmark = pytest.mark.needs_reference
BEV = int(os.environ.get('FIERY_TEST_BEV', '16'))      # cells per side: the distribution encoders halve it four times, and train-mode BatchNorm wants more than a couple of values per channel


@pytest.fixture(scope='module')
def callers():
    from oracle import ref_shims
    return ref_shims.load_reference_callers(batches=[])


def _hparams():
    # (EfficientNet-b0 trunk, two frames in, one out, one GRU block: every branch trainer.py touches, small enough for the simulator)
    cfg = tiny_cfg('baseline.yml', bev=BEV, **{'MODEL.FUTURE_PRED.N_GRU_BLOCKS': 1, 'MODEL.FUTURE_PRED.N_RES_LAYERS': 1,
                                             'N_FUTURE_FRAMES': 1, 'TIME_RECEPTIVE_FIELD': 2, 'MODEL.ENCODER.NAME': 'efficientnet-b0'})
    return cfg.convert_to_dict()


def _batch(cfg, seed, B=2, n=2):
    """What `fiery/data.py:150-239,404-440` yields per sample: images + calibrations + ego-motion + the five label videos."""
    from fiery_amd.synthetic import make_inputs
    S = cfg.TIME_RECEPTIVE_FIELD + cfg.N_FUTURE_FRAMES
    X = int((cfg.LIFT.X_BOUND[1] - cfg.LIFT.X_BOUND[0]) / cfg.LIFT.X_BOUND[2])
    image, K, E, ego = make_inputs(B, S, n, image_hw=tuple(cfg.IMAGE.FINAL_DIM), seed=seed)
    g = torch.Generator().manual_seed(seed + 50)
    inst = torch.zeros(B, S, X, X, dtype=torch.long)
    for t in range(S):
        inst[:, t, 1 + t % 2:4 + t % 2, 2:5] = 1
        inst[:, t, 5:7, 0:2 + t % 2] = 2
        if X >= 16:
            inst[:, t, 9:13, 8 + t:12 + t] = 3
    return dict(image=image, intrinsics=K, extrinsics=E, future_egomotion=0.2 * ego, segmentation=(inst > 0).long().unsqueeze(2),
                instance=inst, centerness=torch.rand(B, S, 1, X, X, generator=g), offset=torch.randn(B, S, 2, X, X, generator=g),
                flow=torch.randn(B, S, 2, X, X, generator=g))


def _modules(callers, sim, hparams):
    """The reference's TrainingModule twice: around its own Fiery, and around fiery_amd's (trainer.py:6 is the one line
    INTEGRATION.md changes - here the module global it binds); same weights in both, through the strict state_dict."""
    from fiery_amd.model import Fiery

    class FieryOnTheSimulator(Fiery):             # the product class + the test hook that points it at the CPU build of its kernels
        def __init__(self, cfg):
            super().__init__(cfg)
            self._lib = sim

    trainer = callers.trainer
    torch.manual_seed(0)
    theirs = trainer.TrainingModule(hparams)
    randomise_weights(theirs.model)
    original = trainer.Fiery
    trainer.Fiery = FieryOnTheSimulator
    try:
        ours = trainer.TrainingModule(hparams)
    finally:
        trainer.Fiery = original
    assert isinstance(ours.model, Fiery) and not isinstance(theirs.model, Fiery)
    with torch.no_grad():
        for name in ('segmentation_weight', 'centerness_weight', 'offset_weight', 'flow_weight'):
            getattr(theirs.model, name).fill_(0.1 + 0.05 * len(name))          # trainer.py:42-64: Parameters attached to the model
    missing = ours.load_state_dict(theirs.state_dict(), strict=True)            # evaluate.py:19's contract, attached weights included
    assert not missing.missing_keys and not missing.unexpected_keys
    return theirs, ours, FieryOnTheSimulator


def _fp64_gradients(module, batch, seed):
    """`shared_step` + backward of a copy of `module` in fp64, with the SAME random draws as the fp32 run: the drop-connect masks
    (`torch.rand(..., dtype=x.dtype)`) and the latent noise (`torch.randn_like(mu)`) are drawn in fp32 from the same generator
    state and widened - an fp64 draw would consume the generator differently and compare two different networks."""
    import copy
    twin = copy.deepcopy(module).double()
    twin.train()
    for sub in twin.modules():                      # plain tensor attributes (losses.py:43's class weights) are not touched by .double()
        for key, value in vars(sub).items():
            if torch.is_tensor(value) and value.dtype == torch.float32:
                setattr(sub, key, value.double())
    rand, randn_like = torch.rand, torch.randn_like

    def rand32(*size, dtype=None, **kw):
        return rand(*size, dtype=torch.float32, **kw).to(dtype or torch.get_default_dtype())

    def randn_like32(x, **kw):
        return torch.randn(x.shape, dtype=torch.float32, device=x.device).to(x.dtype)

    grid_sample = torch.nn.functional.grid_sample

    def grid_sample_any(x, grid, **kw):             # geometry.py:220 hands the sampler `grid.float()` whatever the features' type
        return grid_sample(x, grid.to(x.dtype), **kw)

    torch.rand, torch.randn_like, torch.nn.functional.grid_sample = rand32, randn_like32, grid_sample_any
    torch.set_default_dtype(torch.float64)          # (the reference builds its warping grids and label buffers in the default dtype)
    try:
        torch.manual_seed(seed)
        _, labels, loss = twin.shared_step({k: (v.double() if v.is_floating_point() else v.clone()) for k, v in batch.items()}, True)
        sum(loss.values()).backward()
    finally:
        torch.rand, torch.randn_like, torch.nn.functional.grid_sample = rand, randn_like, grid_sample
        torch.set_default_dtype(torch.float32)
    return {n: p.grad for n, p in twin.named_parameters()}, labels


def test_training_module_shared_step_on_both_classes(callers, sim, monkeypatch):
    """`TrainingModule.shared_step(batch, is_train=True)` (trainer.py:66-131): label warping, forward with the future labels,
    every loss of `fiery/losses.py` incl. the uncertainty weights read from Parameters the trainer attached to the model
    (trainer.py:42-64), then `sum(loss.values()).backward()` as training_step does (:200-208).  Loss terms and every
    parameter's gradient on the product class against the reference class."""
    hparams = _hparams()
    theirs, ours, _ = _modules(callers, sim, hparams)
    batch = _batch(theirs.cfg, seed=3)
    theirs.train()
    ours.train()
    # (train mode draws random numbers - the trunk's drop-connect masks, encoder.py:72-75, and the latent's noise, fiery.py:326-329 -
    # in the same order and shapes on both classes: same seed, same draws)
    torch.manual_seed(11)
    _, labels_t, loss_t = theirs.shared_step({k: v.clone() for k, v in batch.items()}, True)
    torch.manual_seed(11)
    _, labels_o, loss_o = ours.shared_step({k: v.clone() for k, v in batch.items()}, True)
    assert set(loss_t) == set(loss_o) == {'segmentation', 'segmentation_uncertainty', 'instance_center', 'instance_offset',
                                          'centerness_uncertainty', 'offset_uncertainty', 'instance_flow', 'flow_uncertainty',
                                          'probabilistic'}
    for key in labels_t:
        assert torch.equal(labels_t[key], labels_o[key]), key
    for key in loss_t:
        a, b = float(loss_t[key].detach()), float(loss_o[key].detach())
        assert abs(a - b) <= 2e-3 * max(1.0, abs(a)), (key, a, b)            # (train-mode BatchNorm over 2-128 values per channel)
    sum(loss_t.values()).backward()
    sum(loss_o.values()).backward()
    grads_t = {n: p.grad for n, p in theirs.named_parameters()}
    grads_o = {n: p.grad for n, p in ours.named_parameters()}
    assert set(grads_t) == set(grads_o)
    assert [n for n, g in grads_o.items() if (g is None) != (grads_t[n] is None)] == []
    # Every gradient tensor against the reference class's.  The two fp32 evaluations differ by rounding only, and train-mode
    # BatchNorm over 2-128 values per channel magnifies rounding by orders of magnitude - how much depends on the draw, not on
    # the code - so the yardstick is the reference class evaluated in fp64 (same weights, same batch, same random draws: see
    # `_fp64_gradients`): per tensor, the product class's relative L2 distance to the fp64 gradient is at most 3x the reference
    # class's own fp32 distance to it (+1e-3 of the tensor's norm: a tensor the reference happened to round well on).  The
    # biases in front of a train-mode BatchNorm have an analytically zero gradient (what every evaluation computes there is
    # rounding, 1e-9 of the largest gradient in the model): their norms are floored at 1e-5 of that largest gradient.
    # (Until round 5 this was "within 1 % of the fp32 reference": 6e-3 at worst with direct-form convolutions, 1.03e-2 with the
    # Winograd form in the training graph, whose rounding is 2-4x the direct form's - a bound on one rounding draw against
    # another.  The fp64 yardstick bounds what matters: nobody's fp32 arithmetic is much worse than the reference's own.)
    grads_64, labels_64 = _fp64_gradients(theirs, batch, seed=11)
    for key in labels_t:                            # same targets: integer labels equal, warped float labels to fp32 rounding
        if labels_t[key].is_floating_point():
            assert torch.allclose(labels_64[key].float(), labels_t[key], rtol=0, atol=1e-5), key
        else:
            assert torch.equal(labels_64[key], labels_t[key]), key
    assert set(grads_64) == set(grads_t)
    top = max(g.norm().item() for g in grads_t.values() if g is not None)
    rows = []
    for name, gt in grads_t.items():
        if gt is None:
            continue
        go, g64 = grads_o[name], grads_64[name]
        assert torch.isfinite(go).all(), name
        floor = max(g64.norm().item(), 1e-5 * top)
        err_o = (go.double() - g64).norm().item() / floor
        err_t = (gt.double() - g64).norm().item() / floor
        if os.environ.get('FIERY_TEST_VERBOSE'):
            print(f'{name:90s} |g| {gt.norm().item():10.3e}  ours {err_o:9.2e}  reference fp32 {err_t:9.2e}')
        rows.append((err_o / (3 * err_t + 1e-3), err_o, err_t, name))
    rows.sort(reverse=True)
    parity_report.record('reference callers: TrainingModule.shared_step + backward', f'gradients, worst of {len(rows)} tensors ({rows[0][3]})',
                         rows[0][1], 1.0, rows[0][1], rows[0][2], 3 * rows[0][2] + 1e-3,
                         'relative L2 against the fp64 evaluation of the reference class; yardstick = the reference class in fp32')
    assert rows[0][0] <= 1.0, rows[:5]
    assert max(r[1] for r in rows) < 2e-2, max(rows, key=lambda r: r[1])          # and nothing is off by more than rounding can explain
    # The DIRECT-FORM training graph (`train_graph.TRAIN_WINOGRAD = False`, FIERY_TRAIN_WINOGRAD=0) stays gated at the bar this
    # test held before the Winograd form entered the graph: every gradient tensor within 1 % of the reference class's fp32
    # gradient, measured against the tensor's own norm (same floor for the analytically-zero biases).  What the Winograd form
    # costs beyond that is the explicit allowance above - never more than 3x the reference's own rounding - and is recorded
    # in the ledger next to the direct form's figure.
    from fiery_amd import train_graph
    monkeypatch.setattr(train_graph, 'TRAIN_WINOGRAD', False)
    ours.zero_grad(set_to_none=True)
    torch.manual_seed(11)
    _, _, loss_d = ours.shared_step({k: v.clone() for k, v in batch.items()}, True)
    sum(loss_d.values()).backward()
    worst = []
    for name, gt in grads_t.items():
        if gt is None:
            continue
        gd = dict(ours.named_parameters())[name].grad
        worst.append(((gd - gt).norm().item() / max(gt.norm().item(), 1e-5 * top), name))
    worst.sort(reverse=True)
    parity_report.record('reference callers: TrainingModule.shared_step + backward', f'gradients, direct-form graph, worst of {len(worst)} tensors ({worst[0][1]})',
                         worst[0][0], 1.0, None, None, 1e-2, 'relative L2 against the reference class in fp32 (the bar of rounds 1-4)')
    assert worst[0][0] < 1e-2, worst[:5]
    monkeypatch.undo()
    # configure_optimizers (trainer.py:252-258) sees the attached weights through model.parameters(); a step rebuilds the plan
    opt = ours.configure_optimizers()
    n_params = sum(len(g['params']) for g in opt.param_groups)
    assert n_params == len(list(ours.model.parameters())) and any(p is ours.model.flow_weight for g in opt.param_groups for p in g['params'])
    opt.step()
    ours.eval()
    with torch.no_grad():
        again = ours.model(batch['image'], batch['intrinsics'], batch['extrinsics'], batch['future_egomotion'])
    assert all(torch.isfinite(v).all() for v in again.values() if v is not None)
    assert 'model.segmentation_weight' in ours.state_dict()


def test_validation_step_and_the_evaluate_loop_on_both_classes(callers, sim, tmp_path, monkeypatch):
    """`shared_step(batch, is_train=False)` (the IoU / panoptic metric updates of trainer.py:118-129) and the whole of
    `evaluate.eval` - checkpoint loading with strict=True, `model.cfg` mutation (evaluate.py:27-32), the loop body with
    `noise=zeros(B, 1, model.latent_dim)`, `predict_instance_segmentation_and_trajectories(make_consistent=True)`, both metrics
    per evaluation range - executed from the reference's file with its device literal replaced by the CPU and its evaluation
    ranges scaled to the test's grid.  Same numbers from both classes."""
    import io
    import contextlib
    hparams = _hparams()
    theirs, ours, sim_class = _modules(callers, sim, hparams)
    theirs.eval()
    ours.eval()
    batch = _batch(theirs.cfg, seed=5)
    with torch.no_grad():
        out_t, _, loss_t = theirs.shared_step({k: v.clone() for k, v in batch.items()}, False)
        out_o, _, loss_o = ours.shared_step({k: v.clone() for k, v in batch.items()}, False)
    for key in out_t:
        if out_t[key] is None:
            assert out_o[key] is None
            continue
        assert (out_t[key] - out_o[key]).abs().max().item() <= 1e-4 * max(1.0, out_t[key].abs().max().item()), key
    for key in loss_t:
        assert abs(float(loss_t[key]) - float(loss_o[key])) <= 1e-4 * max(1.0, abs(float(loss_t[key]))), key
    for name in ('true_positive', 'false_positive', 'false_negative', 'support'):
        assert torch.equal(getattr(theirs.metric_iou_val, name), getattr(ours.metric_iou_val, name)), name
    for name in ('iou', 'true_positive', 'false_positive', 'false_negative'):
        assert torch.allclose(getattr(theirs.metric_panoptic_val, name), getattr(ours.metric_panoptic_val, name), atol=1e-5), name

    # ---- evaluate.py, whole function ------------------------------------------------------------------------------------
    ckpt = os.path.join(tmp_path, 'fiery.ckpt')
    torch.save({'state_dict': theirs.state_dict(), 'hyper_parameters': hparams}, ckpt)
    source = open(callers.evaluate_path).read()
    assert "torch.device('cuda:0')" in source
    source = source.replace("torch.device('cuda:0')", "torch.device('cpu')")
    results = {}
    for which, fiery_class in (('reference', None), ('fiery_amd', sim_class)):
        batches = [{k: v.clone() for k, v in _batch(theirs.cfg, seed=s).items()} for s in (7, 8)]
        import sys
        sys.modules['fiery.data'].prepare_dataloaders = lambda cfg, b=batches: (None, b)
        scope = {'__name__': 'evaluate_under_test'}
        exec(compile(source, callers.evaluate_path, 'exec'), scope)
        scope['EVALUATION_RANGES'] = {'30x30': (BEV // 4, BEV - BEV // 4), '100x100': (0, BEV)}             # (70, 130) / (0, 200) of a 200-cell grid
        printed = io.StringIO()
        original = callers.trainer.Fiery
        if fiery_class is not None:
            callers.trainer.Fiery = fiery_class
        try:
            with contextlib.redirect_stdout(printed):
                scope['eval'](ckpt, 'no-dataset', 'mini')
        finally:
            callers.trainer.Fiery = original
        lines = printed.getvalue().strip().splitlines()
        results[which] = {lines[i]: [float(x) for x in lines[i + 1].split(' & ')] for i in range(len(lines) - 8, len(lines), 2)}
    assert set(results['reference']) == {'iou', 'pq', 'sq', 'rq'}
    for key, want in results['reference'].items():
        got = results['fiery_amd'][key]
        assert all(abs(a - b) <= 0.1 for a, b in zip(want, got)), (key, want, got)          # printed with one decimal
