"""GPU parity tests of the whole path (network forward, sliding-window predict) against the CPU
oracle and the committed golden fixtures.  Tolerance (BASELINE.json north_star): max-abs 1e-4 on
softmax outputs / logits, identical argmax mask (voxels whose oracle class margin is below the
tolerance are excluded from the mask comparison and counted)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope='module')
def tiny_engine(tiny_net):
    from deepwmh_amd.network import UNetEngine
    plan, net, sd = tiny_net
    return UNetEngine(plan, sd, 'cuda:0')


def _mask_check(seg, seg_ref, prob_ref, tol=2 * TOL):
    sure = np.abs(prob_ref[1] - prob_ref[0]) > tol
    mism = int((seg != seg_ref)[sure].sum())
    return mism, int((~sure).sum())


def test_tiny_net_logits_deep_supervision(tiny_engine, tiny_net, tiny_io):
    plan, net, _ = tiny_net
    x = torch.from_numpy(tiny_io['patch']).cuda()
    ds = tiny_engine.forward_logits(x, deep_supervision=True)
    torch.cuda.synchronize()
    assert len(ds) == plan.num_pool
    with torch.no_grad():
        net.do_ds = True
        ref = net(torch.from_numpy(tiny_io['patch']))
        net.do_ds = False
    for i, t in enumerate(ds):
        got = t.cpu().numpy()
        e_live = float(np.abs(got - ref[i].numpy()).max())
        e_gold = float(np.abs(got - tiny_io['logits_ds%d' % i]).max())
        print('ds%d: max|logit err| live %.3g golden %.3g (|logit| max %.3g)' % (i, e_live, e_gold, np.abs(got).max()))
        assert e_live <= TOL and e_gold <= TOL
    single = tiny_engine.forward_logits(x).cpu().numpy()
    assert np.array_equal(single, ds[0].cpu().numpy())


def test_batch_items_are_independent(tiny_engine, tiny_io):
    # InstanceNorm is per sample: a batch of different patches must equal the per-patch results
    rng = np.random.default_rng(0)
    xs = np.concatenate([tiny_io['patch'], rng.standard_normal((4, 1, 32, 32, 32)).astype(np.float32)], 0)
    all5 = tiny_engine.forward_logits(torch.from_numpy(xs).cuda()).cpu().numpy()
    for b in range(5):
        one = tiny_engine.forward_logits(torch.from_numpy(xs[b:b + 1]).cuda()).cpu().numpy()
        assert np.array_equal(one[0], all5[b]), 'batch item %d differs from its single-item run' % b


@pytest.mark.parametrize('mirror,key', [(False, 'notta'), (True, 'tta')])
def test_tiled_predict_matches_golden(tiny_engine, tiny_io, mirror, key):
    from deepwmh_amd.sliding_window import SlidingWindowPredictor
    from oracle.sliding_window import hard_dice_binary
    pred = SlidingWindowPredictor(tiny_engine, do_mirroring=mirror, batch=8)
    seg, prob = pred.predict(tiny_io['volume'])
    err = float(np.abs(prob - tiny_io['prob_' + key]).max())
    mism, unsure = _mask_check(seg, tiny_io['seg_' + key], tiny_io['prob_' + key])
    dice = hard_dice_binary(tiny_io['seg_' + key], seg)
    print('%s: max|softmax err| %.3g, argmax mismatches %d (+%d voxels inside the tolerance band), dice %.6f'
          % (key, err, mism, unsure, dice))
    assert err <= TOL and mism == 0 and dice > 0.9999
    np.testing.assert_allclose(prob.sum(0), 1.0, atol=1e-5)
    # batch size must not change results (patches are accumulated in a fixed order)
    if not mirror:
        seg3, prob3 = SlidingWindowPredictor(tiny_engine, do_mirroring=False, batch=3).predict(tiny_io['volume'])
        assert np.array_equal(prob3, prob) and np.array_equal(seg3, seg)


@pytest.mark.parametrize('mirror', [False, True])
def test_patch_sharded_backend_is_bit_identical_to_serial(tiny_engine, tiny_io, mirror):
    # the multi-GPU tile/gather/accumulate path (world size 1 here; W > 1 is covered by the gloo CPU tests)
    from deepwmh_amd.distributed import HipTileBackend, PatchShardedPredictor
    from deepwmh_amd.sliding_window import SlidingWindowPredictor
    serial = SlidingWindowPredictor(tiny_engine, do_mirroring=mirror)
    seg, prob = serial.predict(tiny_io['volume'])
    sharded = PatchShardedPredictor(HipTileBackend(SlidingWindowPredictor(tiny_engine, do_mirroring=mirror)))
    seg2, prob2 = sharded.predict(torch.from_numpy(tiny_io['volume']).cuda())
    torch.cuda.synchronize()
    assert np.array_equal(prob2.cpu().numpy(), prob) and np.array_equal(seg2.cpu().numpy(), seg)


def test_predict_volume_smaller_than_patch(tiny_engine, tiny_io):
    from deepwmh_amd.sliding_window import SlidingWindowPredictor
    seg, prob = SlidingWindowPredictor(tiny_engine, do_mirroring=False).predict(tiny_io['small_volume'])
    assert seg.shape == tiny_io['small_volume'].shape[1:]
    assert float(np.abs(prob - tiny_io['prob_small']).max()) <= TOL
    assert _mask_check(seg, tiny_io['seg_small'], tiny_io['prob_small'])[0] == 0


def test_single_tile_volume_uses_unit_weights(tiny_engine, tiny_net, tiny_io):
    # one tile -> nnU-Net skips the Gaussian (weights 1); volume == patch
    from deepwmh_amd.sliding_window import SlidingWindowPredictor
    from oracle import sliding_window as osw
    plan, net, _ = tiny_net
    vol = tiny_io['patch'][0]
    seg, prob = SlidingWindowPredictor(tiny_engine, do_mirroring=False).predict(vol)
    seg_o, prob_o = osw.predict_3d_tiled(net, vol, plan.patch_size, 0.5, False, True)
    assert float(np.abs(prob - prob_o).max()) <= TOL


@pytest.mark.parametrize('shape', [(20, 40, 30), (32, 32, 32), (33, 21, 47), (7, 9, 70)], ids=str)
def test_volumes_smaller_than_or_ragged_against_the_patch(tiny_engine, tiny_net, shape):
    """pad_nd_image edge cases [nnU-Net neural_network.py::_internal_predict_3D_3Dconv_tiled]: axes shorter than the patch are
    zero-padded (lower = diff // 2, upper = the rest) and cropped back; an exact fit is one tile without Gaussian
    weighting; odd sizes give ragged steps.  With and without mirroring, against the CPU oracle."""
    from deepwmh_amd.sliding_window import SlidingWindowPredictor
    from oracle import sliding_window as osw
    plan, net, _ = tiny_net
    vol = osw.synthetic_flair(shape, seed=sum(shape))
    for mirror in (False, True):
        seg, prob = SlidingWindowPredictor(tiny_engine, do_mirroring=mirror).predict(vol)
        seg_o, prob_o = osw.predict_3d_tiled(net, vol, plan.patch_size, 0.5, mirror, True)
        assert prob.shape == prob_o.shape == (2,) + tuple(shape) and seg.shape == tuple(shape)
        err = float(np.abs(prob - prob_o).max())
        mism, unsure = _mask_check(seg, seg_o, prob_o)
        assert err <= TOL and mism == 0, (shape, mirror, err, mism)


def test_mirroring_flip_equivariance(tiny_engine, tiny_io):
    # size-independent property: with all 8 mirrors averaged, predicting a flipped patch-sized volume
    # gives the flipped prediction (single tile, so no Gaussian asymmetry enters)
    from deepwmh_amd.sliding_window import SlidingWindowPredictor
    pred = SlidingWindowPredictor(tiny_engine, do_mirroring=True)
    vol = tiny_io['patch'][0]
    _, p = pred.predict(vol)
    _, pf = pred.predict(np.ascontiguousarray(vol[:, ::-1, :, ::-1]))
    assert float(np.abs(pf[:, ::-1, :, ::-1] - p).max()) <= 2e-5


def test_full_size_patch_forward_vs_oracle():
    """BASELINE.json architecture (31.2 M parameters) on one 128^3 patch against the CPU oracle."""
    import time
    from deepwmh_amd.network import UNetEngine
    from deepwmh_amd.plan import baseline_plan
    from oracle.generic_unet import unet_from_plan, perturb_like_trained, state_dict_numpy
    from oracle.sliding_window import synthetic_flair
    torch.manual_seed(0)
    plan = baseline_plan()
    net = perturb_like_trained(unet_from_plan(plan), seed=2).eval()
    net.do_ds = False
    vol = synthetic_flair((128, 128, 128), seed=1)
    eng = UNetEngine(plan, state_dict_numpy(net), 'cuda:0')
    x = torch.from_numpy(vol[None]).cuda()
    got = eng.forward_logits(x)
    torch.cuda.synchronize()
    t0 = time.time()
    got = eng.forward_logits(x).cpu().numpy()
    t_gpu = time.time() - t0
    t0 = time.time()
    with torch.no_grad():
        ref = net(torch.from_numpy(vol[None])).numpy()
    t_cpu = time.time() - t0
    err = float(np.abs(got - ref).max())
    sm = lambda a: np.exp(a - a.max(1, keepdims=True)) / np.exp(a - a.max(1, keepdims=True)).sum(1, keepdims=True)
    perr = float(np.abs(sm(got) - sm(ref)).max())
    margin = np.abs(ref[:, 1] - ref[:, 0])
    mism = int(((got.argmax(1) != ref.argmax(1)) & (margin > 4 * TOL)).sum())
    print('full-size patch: max|logit err| %.3g, max|softmax err| %.3g, |logit| max %.3g, argmax mismatches %d; '
          'GPU %.3f s (incl. D2H), CPU oracle %.1f s on %d threads'
          % (err, perr, np.abs(ref).max(), mism, t_gpu, t_cpu, torch.get_num_threads()))
    assert perr <= TOL and mism == 0


@pytest.mark.parametrize('patch', [(96, 160, 128), (160, 96, 192), (64, 224, 96)])
def test_full_width_net_non_cubic_patch_vs_oracle(patch):
    """Patch sizes an nnU-Net planner really emits are seldom cubes (`plans_per_stage[...]['patch_size']`, read by
    `UNetPlan.from_nnunet_plans`): the full-width net on one patch of such a plan, default (f16x2) and exact fp32 modes,
    against the CPU oracle.  The rows are 6, 5 and 3 voxels long at the bottleneck; the composed decoder stages, the strided
    pair kernels and the one-role kernels all meet shapes the 128^3 plan never gives them."""
    from deepwmh_amd.network import UNetEngine
    from deepwmh_amd.plan import UNetPlan
    from oracle.generic_unet import unet_from_plan, perturb_like_trained, state_dict_numpy
    from oracle.sliding_window import synthetic_flair
    torch.manual_seed(0)
    plan = UNetPlan(patch_size=list(patch))
    net = perturb_like_trained(unet_from_plan(plan), seed=3).eval()
    net.do_ds = False
    vol = synthetic_flair(patch, seed=4)
    with torch.no_grad():
        ref = net(torch.from_numpy(vol[None])).numpy()
    sm = lambda a: np.exp(a - a.max(1, keepdims=True)) / np.exp(a - a.max(1, keepdims=True)).sum(1, keepdims=True)
    margin = np.abs(ref[:, 1] - ref[:, 0])
    x = torch.from_numpy(vol[None]).cuda()
    for precision in ('f16x2', 'fp32'):
        eng = UNetEngine(plan, state_dict_numpy(net), 'cuda:0', precision=precision)
        got = eng.forward_logits(x).cpu().numpy()
        # a batch of two different patches: items independent of each other and of the batch size
        x2 = torch.cat([x, x.flip(4)])
        got2 = eng.forward_logits(x2).cpu().numpy()
        perr = float(np.abs(sm(got) - sm(ref)).max())
        mism = int(((got.argmax(1) != ref.argmax(1)) & (margin > 4 * TOL)).sum())
        print('patch %s, %s: max|logit err| %.3g, max|softmax err| %.3g, argmax mismatches outside the band %d; batch item 0 '
              'against the single run: max|d| %.3g' % (patch, precision, np.abs(got - ref).max(), perr, mism, np.abs(got2[0] - got[0]).max()))
        assert perr <= TOL and mism == 0
        assert float(np.abs(sm(got2[:1]) - sm(ref)).max()) <= TOL
        del eng


@pytest.fixture(scope='module')
def baseline_pair():
    """BASELINE.json architecture (31.2 M parameters): the oracle net with trained-like parameters and the HIP engine
    holding the same state dict."""
    from deepwmh_amd.network import UNetEngine
    from deepwmh_amd.plan import baseline_plan
    from oracle.generic_unet import unet_from_plan, perturb_like_trained, state_dict_numpy
    torch.manual_seed(0)
    plan = baseline_plan()
    net = perturb_like_trained(unet_from_plan(plan), seed=2).eval()
    net.do_ds = False
    return plan, net, UNetEngine(plan, state_dict_numpy(net), 'cuda:0')


@pytest.mark.timeout(1500)
def test_configs1_full_volume_no_mirroring_vs_oracle(baseline_pair):
    """configs[1] end to end at full size, the workload /root/reference/deepwmh/main/predict.py:153-156 launches (here with
    `--disable_tta`, DCNN_multistage.py:331-336, so that the CPU oracle finishes in about a minute on the GPU box's host):
    160x192x160 volume, 8 tiles of 128^3 at step 0.5, Gaussian aggregation, argmax -- HIP path vs `predict_3d_tiled`."""
    import time
    from deepwmh_amd.sliding_window import SlidingWindowPredictor
    from oracle import sliding_window as osw
    plan, net, eng = baseline_pair
    vol = osw.synthetic_flair((160, 192, 160), seed=3)
    seg, prob = SlidingWindowPredictor(eng, do_mirroring=False, batch=8).predict(vol)
    t0 = time.time()
    tiles = []
    seg_o, prob_o = osw.predict_3d_tiled(net, vol, plan.patch_size, 0.5, False, True, patch_hook=lambda o, p: tiles.append(o))
    t_cpu = time.time() - t0
    assert tiles == [(x, y, z) for x in (0, 32) for y in (0, 64) for z in (0, 32)]
    err = float(np.abs(prob - prob_o).max())
    mism, unsure = _mask_check(seg, seg_o, prob_o)
    dice = osw.hard_dice_binary(seg_o, seg)
    print('configs[1] volume, no mirroring: max|softmax err| %.3g, argmax mismatches %d (+%d voxels inside the %.0e margin band, '
          '%d of them differing), Dice vs oracle mask %.7f, foreground voxels %d; CPU oracle %.1f s on %d threads'
          % (err, mism, unsure, 2 * TOL, int((seg != seg_o).sum()) - mism, dice, int(seg_o.sum()), t_cpu, torch.get_num_threads()))
    assert seg.shape == seg_o.shape == (160, 192, 160)
    assert err <= TOL and mism == 0
    assert int((seg != seg_o).sum()) == 0 or dice > 0.99999
    # the f16x2 mode (products from fp16 pairs, tests/test_gpu_f16x2.py; nnUNet_predict's and bench.py's default): same oracle, gate 5e-5
    from deepwmh_amd.network import UNetEngine
    from oracle.generic_unet import state_dict_numpy
    eng3 = UNetEngine(plan, state_dict_numpy(net), 'cuda:0', precision='f16x2')
    seg3, prob3 = SlidingWindowPredictor(eng3, do_mirroring=False, batch=8).predict(vol)
    err3 = float(np.abs(prob3 - prob_o).max())
    mism3, _ = _mask_check(seg3, seg_o, prob_o)
    print('  f16x2: max|softmax err| %.3g, argmax mismatches %d, Dice vs oracle mask %.7f' % (err3, mism3, osw.hard_dice_binary(seg_o, seg3)))
    assert err3 <= 0.5 * TOL and mism3 == 0


@pytest.mark.timeout(1500)
def test_configs1_full_patch_all_mirrors_vs_oracle(baseline_pair):
    """One 128^3 patch through all 8 mirrored forwards of the 31 M-parameter net against `mirror_and_predict`
    (nnU-Net `_internal_maybe_mirror_and_pred_3D`): the mirror order, the un-flip and the 1/8 averaging at full width."""
    import time
    from deepwmh_amd.sliding_window import SlidingWindowPredictor
    from oracle import sliding_window as osw
    plan, net, eng = baseline_pair
    vol = osw.synthetic_flair((128, 128, 128), seed=4)
    seg, prob = SlidingWindowPredictor(eng, do_mirroring=True, batch=8).predict(vol)       # one tile: no Gaussian weighting
    t0 = time.time()
    ref = osw.mirror_and_predict(net, torch.from_numpy(vol[None]), True, None)[0].numpy()
    t_cpu = time.time() - t0
    seg_o = ref.argmax(0).astype(np.uint8)
    err = float(np.abs(prob - ref).max())
    mism, unsure = _mask_check(seg, seg_o, ref)
    dice = osw.hard_dice_binary(seg_o, seg)
    print('full-width patch, 8 mirrors: max|softmax err| %.3g, argmax mismatches %d (%d voxels in the margin band), Dice %.7f; '
          'CPU oracle %.1f s' % (err, mism, unsure, dice, t_cpu))
    assert err <= TOL and mism == 0 and dice > 0.99999
    # and the mirrored result differs from the un-mirrored one by far more than the tolerance (the test has teeth)
    _, p0 = SlidingWindowPredictor(eng, do_mirroring=False).predict(vol)
    assert float(np.abs(p0 - ref).max()) > 10 * TOL
    # the f16x2 mode against the same oracle output, gate 5e-5
    from deepwmh_amd.network import UNetEngine
    from oracle.generic_unet import state_dict_numpy
    eng3 = UNetEngine(plan, state_dict_numpy(net), 'cuda:0', precision='f16x2')
    seg3, prob3 = SlidingWindowPredictor(eng3, do_mirroring=True, batch=8).predict(vol)
    err3 = float(np.abs(prob3 - ref).max())
    mism3, _ = _mask_check(seg3, seg_o, ref)
    print('  f16x2: max|softmax err| %.3g, argmax mismatches %d, Dice %.7f' % (err3, mism3, osw.hard_dice_binary(seg_o, seg3)))
    assert err3 <= 0.5 * TOL and mism3 == 0


def _fullsize_fixture_check(name, kind, engines):
    """The whole mirrored volume against the oracle's output computed once in the build container (tests/golden/<name>, made by
    tests/golden/make_golden_fullsize.py): lattice + margin-band probabilities, label map outside the 2e-4 band, checksum."""
    import os
    from oracle import sliding_window as osw
    from deepwmh_amd.sliding_window import SlidingWindowPredictor
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', name))
    shape = tuple(int(v) for v in G['shape'])
    n = int(np.prod(shape))
    vol = osw.synthetic_flair(shape, seed=3)
    assert abs(float(np.abs(vol.astype(np.float64)).sum()) - float(G['volume_abs_sum'])) <= 1e-7 * float(G['volume_abs_sum'])
    mask_o = np.unpackbits(G['mask_bits'])[:n].reshape(shape)
    assert int(mask_o.sum()) == int(G['foreground'])
    band = G['band_index'].astype(np.int64)
    band_margin = np.abs(2.0 * G['band_p1'].astype(np.float64) - 1.0)          # p0 = 1 - p1
    out = {}
    for tag, eng in engines:
        seg, prob = SlidingWindowPredictor(eng, do_mirroring=True, batch=8).predict(vol)
        e_lat = float(np.abs(prob[1][2::4, 2::4, 2::4] - G['p1_lattice']).max())
        e_band = float(np.abs(prob[1].reshape(-1)[band] - G['band_p1']).max()) if len(band) else 0.0
        unsure = np.zeros(n, bool)
        unsure[band[band_margin <= 2 * TOL]] = True                            # the same margin band as _mask_check
        diff = (seg.reshape(-1) != mask_o.reshape(-1))
        mism, any_margin = int((diff & ~unsure).sum()), int(diff.sum())
        mean_err = abs(float(prob[1].astype(np.float64).sum()) - float(G['sum_p1'])) / n
        dice = osw.hard_dice_binary(mask_o, seg)
        print('configs[1] WITH mirroring (%s net), %s vs the oracle fixture: max|softmax err| lattice %.3g, margin band (%d voxels) %.3g; '
              'argmax mismatches %d outside the %.0e band (%d anywhere); |mean p1 - oracle| %.3g; Dice %.7f; foreground %d'
              % (kind, tag, e_lat, len(band), e_band, mism, 2 * TOL, any_margin, mean_err, dice, int(mask_o.sum())))
        out[tag] = (max(e_lat, e_band), mism, mean_err, dice)
    return out


def _state_abs_sum(net):
    return float(sum(float(v.double().abs().sum()) for v in net.state_dict().values()))


def test_configs1_full_volume_with_mirroring_vs_oracle_fixture(baseline_pair):
    """configs[1] exactly as `DeepWMH_predict` runs it (/root/reference/deepwmh/main/predict.py:153-156: mirroring on): the whole
    160x192x160 volume, 8 tiles x 8 mirrored forwards of the 31 M-parameter net, Gaussian aggregation, argmax (VERDICT r02 item 5a;
    before, this combination was only checked at 40x48x40).  Exact-fp32 path (gate 1e-4) and the f16x2 mode (gate 5e-5)."""
    import os
    from deepwmh_amd.network import UNetEngine
    from oracle.generic_unet import state_dict_numpy
    plan, net, eng = baseline_pair
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'fullsize_tta.npz'))
    assert abs(_state_abs_sum(net) - float(G['state_abs_sum'])) <= 1e-7 * float(G['state_abs_sum']), 'not the weights the fixture was made with'
    eng3 = UNetEngine(plan, state_dict_numpy(net), 'cuda:0', precision='f16x2')
    res = _fullsize_fixture_check('fullsize_tta.npz', 'perturbed', [('fp32', eng), ('f16x2', eng3)])
    for tag, (err, mism, mean_err, dice) in res.items():
        assert err <= (TOL if tag == 'fp32' else 0.5 * TOL) and mism == 0 and mean_err <= 5e-6 and dice > 0.9999, tag


def test_configs1_full_volume_with_mirroring_he_init_net_vs_oracle_fixture():
    """The same on the plain He-init network (InitWeights_He(1e-2), biases 0, InstanceNorm affine (1, 0)): its probabilities hug 0.5
    (28 k foreground voxels, 2.5 k voxels inside the 1e-3 margin band) -- the kind of net on which round 2's bf16-pair mode measured
    6.1e-5.  Fixture: tests/golden/fullsize_tta_he.npz (make_golden_fullsize.py --net he)."""
    import os
    from deepwmh_amd.network import UNetEngine
    from deepwmh_amd.plan import baseline_plan
    from oracle.generic_unet import state_dict_numpy, unet_from_plan
    torch.manual_seed(0)
    plan = baseline_plan()
    net = unet_from_plan(plan).eval()
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'fullsize_tta_he.npz'))
    assert abs(_state_abs_sum(net) - float(G['state_abs_sum'])) <= 1e-7 * float(G['state_abs_sum']), 'not the weights the fixture was made with'
    sd = state_dict_numpy(net)
    res = _fullsize_fixture_check('fullsize_tta_he.npz', 'He-init', [('fp32', UNetEngine(plan, sd, 'cuda:0')),
                                                                    ('f16x2', UNetEngine(plan, sd, 'cuda:0', precision='f16x2'))])
    for tag, (err, mism, mean_err, dice) in res.items():
        assert err <= (TOL if tag == 'fp32' else 0.5 * TOL) and mism == 0 and mean_err <= 5e-6 and dice > 0.9999, tag



def test_full_size_volume_properties():
    """BASELINE.json configs[1] at full size (160x192x160, 128^3 patches, mirroring): size-independent properties, since the
    CPU oracle needs minutes for this volume.  (1) determinism: two runs give the same bits; (2) every voxel's class
    probabilities sum to 1 and the label map is their argmax; (3) the patch-sharded code path on one rank (tile by tile,
    HipTileBackend) gives the bits of the serial path; (4) without mirroring, batching the 8 patches into one forward or
    running them one at a time agrees to rounding level (samples are independent; the launch shapes differ)."""
    from deepwmh_amd.distributed import HipTileBackend, PatchShardedPredictor
    from deepwmh_amd.network import UNetEngine
    from deepwmh_amd.plan import baseline_plan
    from deepwmh_amd.sliding_window import SlidingWindowPredictor
    from deepwmh_amd.synthetic import he_init_state_dict, synthetic_flair
    plan = baseline_plan()
    eng = UNetEngine(plan, he_init_state_dict(plan, seed=0), 'cuda:0')
    vol = torch.from_numpy(synthetic_flair((160, 192, 160), seed=0)).cuda()
    pred = SlidingWindowPredictor(eng, do_mirroring=True, batch=8)
    seg, prob = pred.predict_device(vol)
    seg, prob = seg.clone(), prob.clone()
    seg2, prob2 = pred.predict_device(vol)
    assert torch.equal(seg, seg2) and torch.equal(prob, prob2)
    assert tuple(prob.shape) == (2, 160, 192, 160) and tuple(seg.shape) == (160, 192, 160)
    assert float((prob.sum(0) - 1).abs().max()) <= 1e-6
    assert float(prob.min()) >= 0.0 and float(prob.max()) <= 1.0
    assert torch.equal(seg.long(), prob.argmax(0))
    seg3, prob3 = PatchShardedPredictor(HipTileBackend(pred)).predict(vol)
    assert torch.equal(seg, seg3) and torch.equal(prob, prob3)
    p8 = SlidingWindowPredictor(eng, do_mirroring=False, batch=8)
    p1 = SlidingWindowPredictor(eng, do_mirroring=False, batch=1)
    s8, q8 = p8.predict_device(vol)
    s8, q8 = s8.clone(), q8.clone()
    s1, q1 = p1.predict_device(vol)
    # not bit-identical: the batch size moves the small layers between the split-K and the unsplit launch (different
    # summation order); the difference stays at rounding level
    d = float((q8 - q1).abs().max())
    flips = int(((s8 != s1) & ((q8[1] - 0.5).abs() > 1e-4)).sum())
    print('full-size volume, batch 8 vs batch 1 forwards: max |dP| %.3g, label changes outside near-ties %d' % (d, flips))
    assert d <= 2e-5 and flips == 0


def test_bf16_forward_config3(tiny_net, tiny_io):
    """BASELINE.json config 3: bf16 matrix-core operands.  Not a 1e-4 path: report the distance to fp32 and the Dice."""
    from deepwmh_amd.network import UNetEngine
    from deepwmh_amd.sliding_window import SlidingWindowPredictor
    from oracle.sliding_window import hard_dice_binary
    plan, net, sd = tiny_net
    eng = UNetEngine(plan, sd, 'cuda:0', precision='bf16')
    seg, prob = SlidingWindowPredictor(eng, do_mirroring=False).predict(tiny_io['volume'])
    err = float(np.abs(prob - tiny_io['prob_notta']).max())
    dice = hard_dice_binary(tiny_io['seg_notta'], seg)
    print('bf16 operands: max|softmax - fp32 oracle| %.3g, Dice vs fp32 mask %.5f' % (err, dice))
    assert err < 0.05 and dice > 0.98


def test_anisotropic_plan_matches_oracle():
    """nnU-Net plans for thick-slice data: [1,3,3] kernels and [1,2,2] pooling in the first stages."""
    from deepwmh_amd.network import UNetEngine
    from deepwmh_amd.plan import UNetPlan
    from deepwmh_amd.sliding_window import SlidingWindowPredictor
    from oracle import sliding_window as osw
    from oracle.generic_unet import unet_from_plan, perturb_like_trained, state_dict_numpy
    torch.manual_seed(4)
    plan = UNetPlan(1, 8, 2, [[1, 2, 2], [1, 2, 2], [2, 2, 2]], [[1, 3, 3], [1, 3, 3], [3, 3, 3], [3, 3, 3]], 2, 64, [8, 32, 48])
    net = perturb_like_trained(unet_from_plan(plan), seed=5).eval()
    eng = UNetEngine(plan, state_dict_numpy(net), 'cuda:0')
    vol = osw.synthetic_flair((12, 40, 64), seed=9)
    x = torch.from_numpy(np.ascontiguousarray(vol[None, :, 2:10, 4:36, 8:56]))
    with torch.no_grad():
        net.do_ds = True
        ref = net(x)
        net.do_ds = False
    got = eng.forward_logits(x.cuda(), deep_supervision=True)
    for i, (a, b) in enumerate(zip(got, ref)):
        assert tuple(a.shape) == tuple(b.shape)
        assert float((a.cpu() - b).abs().max()) <= TOL, 'scale %d' % i
    seg, prob = SlidingWindowPredictor(eng, do_mirroring=True).predict(vol)
    seg_o, prob_o = osw.predict_3d_tiled(net, vol, plan.patch_size, 0.5, True, True)
    assert float(np.abs(prob - prob_o).max()) <= TOL
    assert _mask_check(seg, seg_o, prob_o)[0] == 0


@pytest.mark.parametrize('precision', ['fp32', 'f16x2'])
def test_two_modalities_three_classes_vs_oracle(precision):
    """Outside DeepWMH's own configuration (one FLAIR channel, lesion / background) but inside the kernels' repertoire: two input modalities
    (the first layer's two-channel form) and three classes (head, softmax, argmax, loss and online evaluation with K = 3) -- forward with deep
    supervision, tiled prediction with mirroring, and one training step's loss and gradients against the oracle."""
    from deepwmh_amd.network import UNetEngine
    from deepwmh_amd.plan import UNetPlan
    from deepwmh_amd.sliding_window import SlidingWindowPredictor
    from deepwmh_amd.trainer import UNetTrainer, downsample_targets
    from oracle import sliding_window as osw, training as ot
    from oracle.generic_unet import unet_from_plan, perturb_like_trained, state_dict_numpy
    torch.manual_seed(11)
    plan = UNetPlan(2, 8, 3, [[2, 2, 2]] * 3, [[3, 3, 3]] * 4, 2, 64, [32, 32, 32])
    net = perturb_like_trained(unet_from_plan(plan), seed=6).eval()
    sd = state_dict_numpy(net)
    eng = UNetEngine(plan, sd, 'cuda:0', precision=precision)
    vol = np.concatenate([osw.synthetic_flair((40, 48, 40), seed=21), osw.synthetic_flair((40, 48, 40), seed=22)], 0)
    x = torch.from_numpy(np.ascontiguousarray(vol[None, :, 4:36, 8:40, 2:34]))
    with torch.no_grad():
        net.do_ds = True
        ref = net(x)
        net.do_ds = False
    got = eng.forward_logits(x.cuda(), deep_supervision=True)
    for i, (a, b) in enumerate(zip(got, ref)):
        assert tuple(a.shape) == tuple(b.shape) and a.shape[1] == 3
        assert float((a.cpu() - b).abs().max()) <= TOL, 'scale %d' % i
    seg, prob = SlidingWindowPredictor(eng, do_mirroring=True).predict(vol)
    seg_o, prob_o = osw.predict_3d_tiled(net, vol, plan.patch_size, 0.5, True, True)
    assert prob.shape[0] == 3 and float(np.abs(prob - prob_o).max()) <= TOL
    top2 = np.sort(prob_o, 0)
    sure = (top2[-1] - top2[-2]) > 2 * TOL
    assert int((seg != seg_o)[sure].sum()) == 0 and set(np.unique(seg_o)) >= {0, 1}
    # one training step: loss and gradients against float64 autograd, online-evaluation counts against the oracle's
    import copy
    data = np.stack([vol[:, 4:36, 8:40, 2:34], vol[:, 8:40, 0:32, 6:38]]).astype(np.float32)
    labels = np.stack([seg_o[4:36, 8:40, 2:34], (seg_o[8:40, 0:32, 6:38] + 1) % 3]).astype(np.uint8)
    n64 = copy.deepcopy(net).double().train()
    n64.do_ds = True
    tgt = ot.downsample_targets(torch.from_numpy(labels[:, None].astype(np.float32)), plan.num_pool)
    out64 = n64(torch.from_numpy(data).double())
    loss = ot.multiple_output_loss(out64, [t.double() for t in tgt], ot.deep_supervision_weights(plan.num_pool))
    loss.backward()
    tr = UNetTrainer(plan, sd, 'cuda:0', precision=precision)
    td = downsample_targets(torch.from_numpy(labels).cuda(), plan.num_pool)
    tr.forward(torch.from_numpy(data).cuda())
    l = float(tr.backward(td).cpu())
    assert abs(l - float(loss.detach())) <= 2e-5 * (1 + abs(float(loss.detach())))
    num = den = 0.0
    for name, p in n64.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        num += float(((tr.g[name].double().cpu() - g) ** 2).sum()); den += float((g ** 2).sum())
    assert (num / den) ** 0.5 <= 2e-3, (num / den) ** 0.5
    counts = tr.online_eval(td[0]).cpu().numpy()
    tp, fp, fn = ot.run_online_evaluation(out64[0].detach().float(), torch.from_numpy(labels[:, None].astype(np.float32)))
    want = np.concatenate([tp, fp, fn]).astype(np.int64)
    assert np.abs(counts - want).sum() <= 4, (counts, want)         # (argmax of logits that differ by rounding: a handful of voxels at most)
