"""CPU tests: the oracle against the committed golden fixtures, and the product's host logic
(steps, Gaussian map, plan) against the oracle."""
import json
import os

import numpy as np
import torch

from deepwmh_amd import sliding_window as psw
from deepwmh_amd.plan import UNetPlan, baseline_plan, tiny_plan
from oracle import sliding_window as osw
from oracle.generic_unet import unet_from_plan


def test_steps_table_golden(golden_dir):
    table = json.load(open(os.path.join(golden_dir, 'steps_table.json')))
    assert len(table) > 20
    for row in table:
        exp = row['steps']
        assert osw.compute_steps_for_sliding_window(row['patch'], row['image'], row['step']) == exp
        assert psw.compute_steps_for_sliding_window(row['patch'], row['image'], row['step']) == exp


def test_steps_baseline_config():
    # (160,192,160) @ 128^3 step 0.5 -> x in {0,32}, y in {0,64}, z in {0,32}
    assert psw.compute_steps_for_sliding_window((128,) * 3, (160, 192, 160), 0.5) == [[0, 32], [0, 64], [0, 32]]


def test_steps_errors():
    import pytest
    with pytest.raises(ValueError):
        psw.compute_steps_for_sliding_window((128,) * 3, (100, 128, 128), 0.5)
    with pytest.raises(ValueError):
        psw.compute_steps_for_sliding_window((32,) * 3, (40, 40, 40), 0.0)


def test_gaussian_golden_and_product_bit_exact(golden_dir):
    g = np.load(os.path.join(golden_dir, 'gaussian.npz'))
    g32 = osw.get_gaussian((32, 32, 32))
    assert np.array_equal(g32, g['g32'])
    assert np.array_equal(psw.get_gaussian((32, 32, 32)), g['g32'])      # product restatement == scipy path
    g128 = psw.get_gaussian((128, 128, 128))
    for idx, val in zip(g['g128_probe_idx'], g['g128_probe_val']):
        assert g128[tuple(idx)] == val
    assert g128.min() == g['g128_min'] and g128.max() == g['g128_max'] == 1.0
    assert abs(float(g128.astype(np.float64).sum()) - float(g['g128_sum'])) < 1e-6
    for ps in [(16, 24, 40), (20, 20, 20), (48, 64, 32)]:
        assert np.array_equal(psw.get_gaussian(ps), osw.get_gaussian(ps))


def test_oracle_forward_matches_golden(tiny_net, tiny_io):
    plan, net, _ = tiny_net
    with torch.no_grad():
        net.do_ds = True
        ds = net(torch.from_numpy(tiny_io['patch']))
    assert len(ds) == plan.num_pool
    for i, t in enumerate(ds):
        np.testing.assert_allclose(t.numpy(), tiny_io['logits_ds%d' % i], rtol=0, atol=2e-5)
    assert ds[0].shape == (1, 2, 32, 32, 32) and ds[1].shape == (1, 2, 16, 16, 16)


def test_oracle_tiled_predict_matches_golden(tiny_net, tiny_io):
    plan, net, _ = tiny_net
    seg, prob = osw.predict_3d_tiled(net, tiny_io['volume'], plan.patch_size, 0.5, False, True)
    np.testing.assert_allclose(prob, tiny_io['prob_notta'], rtol=0, atol=2e-5)
    assert (seg != tiny_io['seg_notta']).mean() < 1e-3
    np.testing.assert_allclose(prob.sum(0), 1.0, atol=1e-5)
    seg_s, prob_s = osw.predict_3d_tiled(net, tiny_io['small_volume'], plan.patch_size, 0.5, False, True)
    assert seg_s.shape == tiny_io['small_volume'].shape[1:]
    np.testing.assert_allclose(prob_s, tiny_io['prob_small'], rtol=0, atol=2e-5)


def test_oracle_mirroring_is_flip_equivariant(tiny_net, tiny_io):
    # size-independent property of the TTA path: mean over all 8 flips commutes with flipping the input
    plan, net, _ = tiny_net
    x = torch.from_numpy(tiny_io['patch'])
    a = osw.mirror_and_predict(net, x, True, None)
    b = osw.mirror_and_predict(net, torch.flip(x, (2, 4)), True, None)
    np.testing.assert_allclose(torch.flip(b, (2, 4)).numpy(), a.numpy(), atol=1e-5)


def test_pad_nd_image_centred():
    v = np.ones((1, 28, 48, 31), np.float32)
    p, sl = osw.pad_nd_image(v, (32, 32, 32))
    assert p.shape == (1, 32, 48, 32)
    assert sl[1] == slice(2, 30) and sl[2] == slice(0, 48) and sl[3] == slice(0, 31)
    assert p[0, :2].sum() == 0 and p[0, 30:].sum() == 0 and p[0, :, :, 31].sum() == 0


def test_hard_dice_binary():
    a = np.zeros((4, 4, 4), np.float32); b = a.copy()
    a[:2] = 1; b[1:3] = 1
    assert abs(osw.hard_dice_binary(a, b) - 2 * 16 / (32 + 32 + 1e-6)) < 1e-9
    assert osw.hard_dice_binary(a, a) > 0.999999


def test_plan_baseline_architecture():
    p = baseline_plan()
    p.check_supported()
    assert p.features() == [32, 64, 128, 256, 320, 320]
    enc = p.encoder_convs()
    assert [(c.cin, c.cout, c.stride) for c in enc[:4]] == [(1, 32, 1), (32, 32, 1), (32, 64, 2), (64, 64, 1)]
    assert enc[-2].name == 'conv_blocks_context.5.0.blocks.0' and (enc[-2].cin, enc[-2].cout, enc[-2].stride) == (320, 320, 2)
    dec = p.decoder_stages()
    assert dec[0]['tu'] == ('tu.0', 320, 320) and dec[0]['convs'][0].cin == 640
    assert dec[-1]['tu'] == ('tu.4', 64, 32) and dec[-1]['convs'][0].cin == 64 and dec[-1]['seg'] == 'seg_outputs.4'
    # every key the plan names exists in the oracle's state dict with the plan's shape (31.19 M params)
    net = unet_from_plan(p)
    sd = net.state_dict()
    for c in enc + [c for d in dec for c in d['convs']]:
        assert tuple(sd[c.name + '.conv.weight'].shape) == (c.cout, c.cin, 3, 3, 3)
        assert c.name + '.instnorm.weight' in sd
    assert abs(sum(v.numel() for v in net.parameters()) / 1e6 - 31.19) < 0.02


def test_plan_from_nnunet_plans_dict():
    plans = dict(num_modalities=1, base_num_features=32, num_classes=1, conv_per_stage=2,
                 plans_per_stage={0: dict(patch_size=np.array([128, 128, 128]), pool_op_kernel_sizes=[[2, 2, 2]] * 5,
                                          conv_kernel_sizes=[[3, 3, 3]] * 6, batch_size=2)})
    p = UNetPlan.from_nnunet_plans(plans)
    assert p.num_classes == 2 and p.patch_size == [128, 128, 128] and p.num_pool == 5
    import pytest
    aniso = tiny_plan(); aniso.conv_kernel_sizes = [[1, 3, 3]] + [[3, 3, 3]] * 3; aniso.pool_op_kernel_sizes = [[1, 2, 2], [2, 2, 2], [2, 2, 2]]
    aniso.patch_size = [8, 32, 32]
    aniso.check_supported()
    assert not aniso.isotropic
    e = aniso.encoder_convs()
    assert (e[0].kz, e[0].sz, e[0].stride) == (1, 1, 1) and (e[2].kz, e[2].sz, e[2].stride) == (3, 1, 2) and (e[4].sz, e[4].stride) == (2, 2)
    d = aniso.decoder_stages()
    assert d[-1]['tu_kt'] == 1 and d[0]['tu_kt'] == 2
    bad = tiny_plan(); bad.conv_kernel_sizes = [[3, 1, 3]] + [[3, 3, 3]] * 3
    with pytest.raises(NotImplementedError):
        bad.check_supported()


def test_trained_tiny_fixture_is_what_the_oracle_computes(tiny_net, tiny_io, golden_dir):
    """tests/golden/tiny_net_trained20.npz (the "trained-like" variant, 20 oracle SGD steps): the oracle's training step
    reproduces the stored loss sequence and parameters, and the oracle's tiled predictor the stored probabilities of the trained net."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('make_golden_trained_tiny', os.path.join(golden_dir, 'make_golden_trained_tiny.py'))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    G = np.load(os.path.join(golden_dir, 'tiny_net_trained20.npz'))
    plan, _, sd0 = tiny_net
    vol = tiny_io['volume'][0]
    lab = (vol > float(G['threshold'])).astype(np.uint8)
    assert int(lab.sum()) == int(G['label_voxels']) and np.array_equal(mk.schedule(), G['schedule'])
    net, losses = mk.train_oracle(plan, sd0, vol, lab, G['schedule'])
    # (torch's CPU kernels sum in an order that depends on the thread count: tolerance, not bits)
    np.testing.assert_allclose(losses, G['losses'], rtol=0, atol=2e-4)
    assert losses[-1] < 0.25 * losses[0]                                  # it has learnt something
    for k, v in net.state_dict().items():
        np.testing.assert_allclose(v.numpy(), G['w.' + k], rtol=0, atol=2e-4, err_msg=k)
    # the stored weights through the predictor
    net2 = unet_from_plan(plan).eval()
    net2.load_state_dict({k[2:]: torch.from_numpy(G[k]) for k in G.files if k.startswith('w.')}, strict=True)
    seg, prob = osw.predict_3d_tiled(net2, tiny_io['volume'], plan.patch_size, 0.5, False, True)
    assert float(np.abs(prob[1] - G['prob1_notta']).max()) <= 2e-5
    sure = np.abs(2.0 * G['prob1_notta'] - 1.0) > 2e-4
    assert np.array_equal(seg[sure], G['seg_notta'][sure])
    # most voxels are decided after 20 steps (median |p - 0.5| 0.30), unlike the random-init fixtures whose probabilities hug 0.5
    assert float((np.abs(G['prob1_tta'] - 0.5) > 0.25).mean()) > 0.5
