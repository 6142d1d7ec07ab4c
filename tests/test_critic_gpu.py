"""GPU parity of the critic (forward, WGAN-GP loss and gradient, input gradient) and of the optimizer
kernels against oracle/gan_torch.py.  fp32-MFMA path: tight; bf16-MFMA path: bf16 tolerance."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import gan_torch as og

pytestmark = pytest.mark.gpu


def _setup(batch, nx, layers, seed):
    from tc_gan_amd.critic import Critic
    rs = np.random.RandomState(seed)
    c32 = Critic(nx, layers, seed=seed, precision='fp32')
    params_o = [og.t64(p) for p in c32.get_param_values()]
    xg = rs.rand(batch, nx) * 5
    xd = rs.rand(batch, nx) * 5
    eps = rs.rand(batch, 1)
    xp = eps * xd + (1 - eps) * xg
    cond = np.stack([np.full(batch, 20.), rs.rand(batch) * 2 - 1, rs.randint(0, 2, batch)], axis=1)
    return c32, params_o, xg, xd, xp, cond


@pytest.mark.parametrize('batch,nx,layers', [(7, 4, [9]), (64, 8, [32, 32]), (130, 8, [128, 128, 128]),
                                              (1024, 8, [512, 512, 512]), (33, 5, [])])
def test_critic_loss_and_gradient_fp32(batch, nx, layers):
    c, params_o, xg, xd, xp, cond = _setup(batch, nx, layers, seed=batch)
    ps = [p.clone().requires_grad_(True) for p in params_o]
    tg, td, tp, tc = (og.t64(a) for a in (xg, xd, xp, cond))
    loss_o = og.critic_loss(ps, tg, td, tp, tc, tc, tc, 10.0)
    grads_o = torch.autograd.grad(loss_o, ps)
    flat_o = np.concatenate([g.numpy().ravel() for g in grads_o])
    stats = c.loss_grad(xg, cond, xd, cond, xp, cond, 10.0).cpu().numpy()
    np.testing.assert_allclose(stats[3], float(loss_o), rtol=2e-5, atol=1e-5)
    got = c.grads.cpu().numpy()
    np.testing.assert_allclose(got, flat_o, rtol=1e-3, atol=2e-5 * np.abs(flat_o).max())
    d = c.forward(xg, cond).cpu().numpy()
    np.testing.assert_allclose(d, og.critic_forward(params_o, tg, tc)[:, 0].numpy(), rtol=1e-4, atol=1e-5)
    assert abs(c.accuracy(xg, cond, xd, cond) - (stats[0] - stats[1])) < 1e-5


@pytest.mark.parametrize('nonlinearity', ['leaky_rectify', 'very_leaky_rectify', 'linear'])
@pytest.mark.parametrize('batch,nx,layers', [(7, 4, [9]), (130, 8, [128, 128, 128]), (256, 8, [512, 512, 512])])
def test_piecewise_linear_nonlinearities(nonlinearity, batch, nx, layers):
    """simple_discriminator.py:139-152 takes any `lasagne.nonlinearities` name for the hidden layers (CLI
    `--disc-nonlinearity`); the piecewise-linear ones (LeakyRectify(0.01), LeakyRectify(1/3), identity) keep the input gradient a
    linear chain with fixed slopes, so loss, WGAN-GP double backward, critic values and the generator-side input gradient run
    on the same kernels with a slope.  Against torch autograd on the fp64 restatement, like the rectify cases."""
    from tc_gan_amd.critic import Critic
    rs = np.random.RandomState(batch + len(nonlinearity))
    c = Critic(nx, layers, seed=batch, precision='fp32', nonlinearity=nonlinearity)
    params_o = [og.t64(p) for p in c.get_param_values()]
    xg, xd = rs.rand(batch, nx) * 5, rs.rand(batch, nx) * 5
    eps = rs.rand(batch, 1)
    xp = eps * xd + (1 - eps) * xg
    cond = np.stack([np.full(batch, 20.), rs.rand(batch) * 2 - 1, rs.randint(0, 2, batch)], axis=1)
    ps = [p.clone().requires_grad_(True) for p in params_o]
    tg, td, tp, tc = (og.t64(a) for a in (xg, xd, xp, cond))
    loss_o = og.critic_loss(ps, tg, td, tp, tc, tc, tc, 10.0, nonlinearity=nonlinearity)
    flat_o = np.concatenate([g.numpy().ravel() for g in torch.autograd.grad(loss_o, ps)])
    stats = c.loss_grad(xg, cond, xd, cond, xp, cond, 10.0).cpu().numpy()
    np.testing.assert_allclose(stats[3], float(loss_o), rtol=2e-5, atol=1e-5)
    np.testing.assert_allclose(c.grads.cpu().numpy(), flat_o, rtol=1e-3, atol=2e-5 * np.abs(flat_o).max())
    np.testing.assert_allclose(c.forward(xg, cond).cpu().numpy(),
                               og.critic_forward(params_o, tg, tc, nonlinearity=nonlinearity)[:, 0].numpy(), rtol=1e-4, atol=1e-5)
    x = tg.clone().requires_grad_(True)
    want = torch.autograd.grad(-og.critic_forward(params_o, x, tc, nonlinearity=nonlinearity).mean(), x)[0].numpy()
    gx, _ = c.input_grad(xg, cond, -1.0 / batch)
    np.testing.assert_allclose(gx.cpu().numpy(), want, rtol=1e-3, atol=2e-5 * np.abs(want).max())
    with pytest.raises(NotImplementedError):
        Critic(nx, layers, nonlinearity='softmax')            # (a name of lasagne.nonlinearities that is not a hidden-layer choice here)


def _assert_close(got, want, rtol, atol, kinked=False, err_msg='', loose=None):
    """allclose; for a nonlinearity whose f'' jumps (elu at 0: 1 -> 0) a unit whose pre-activation rounds to the other side of
    the kink in fp32 moves one sample's share of a whole weight column, so there a few thousandths of the elements may miss the
    tight bound -- by no more than 1e-3 of the largest element."""
    if not kinked:
        np.testing.assert_allclose(got, want, rtol=rtol, atol=atol, err_msg=err_msg)
        return
    bad = np.abs(got - want) > atol + rtol * np.abs(want)
    assert bad.mean() <= 4e-3, (err_msg, bad.mean())
    np.testing.assert_allclose(got, want, rtol=rtol, atol=max(atol, 1e-3 * np.abs(want).max(), loose or 0.0), err_msg=err_msg)


def _general_case(nonlinearity, norm, batch, nx, layers, precision='fp32', net_options=None, use_scale='auto', conditional=True, seed=0):
    from tc_gan_amd.critic import Critic
    rs = np.random.RandomState(batch + len(nonlinearity) + seed)
    c = Critic(nx, layers, seed=batch, precision=precision, nonlinearity=nonlinearity, normalization=norm,
               net_options=net_options, conditional=conditional)
    # scales away from their initial 1 and biases away from ~0: every term of the double backward carries weight
    flat = c.get_flat()
    off = 0
    for kind, shape in c.param_shapes():
        n = int(np.prod(shape))
        if kind == 'scales':
            flat[off:off + n] = rs.uniform(0.5, 1.5, n)
        elif kind == 'b':
            flat[off:off + n] = rs.normal(0, 0.3, n)
        off += n
    c.set_flat(flat)
    params_o = [og.t64(p) for p in c.get_param_values()]
    xg, xd = rs.rand(batch, nx) * 5, rs.rand(batch, nx) * 5
    eps = rs.rand(batch, 1)
    xp = eps * xd + (1 - eps) * xg
    cond = np.stack([np.full(batch, 20.), rs.rand(batch) * 2 - 1, rs.randint(0, 2, batch)], axis=1) if conditional else None
    kw = dict(nonlinearity=nonlinearity, normalization=norm, use_scale=use_scale)
    return c, params_o, xg, xd, xp, cond, kw


@pytest.mark.parametrize('nonlinearity', ['tanh', 'sigmoid', 'softplus', 'elu'])
@pytest.mark.parametrize('norm', ['none', 'layer', ['none', 'layer', 'layer']])
@pytest.mark.parametrize('batch,nx,layers', [(7, 4, [9, 6, 5]), (130, 8, [128, 128, 128]), (256, 8, [512, 512, 512])])
def test_smooth_nonlinearities_and_scale_layers_vs_fp64_autograd(nonlinearity, norm, batch, nx, layers):
    """`--disc-nonlinearity tanh / sigmoid / softplus / elu` (simple_discriminator.py:139-152) with and without layer
    normalisation -- whose layers then carry the reference's learnable ScaleLayer (:57-75) -- against torch autograd on the
    fp64 restatement: loss, EVERY parameter gradient through the gradient penalty (the curvature f'' of the nonlinearity at
    every layer, the scales, the normalisation), critic values, the generator-side input gradient, the accuracy.  Same
    tolerances as the rectify cases: 2e-5 on the loss, 1e-3 / 2e-5 of the largest element on gradients."""
    c, params_o, xg, xd, xp, cond, kw = _general_case(nonlinearity, norm, batch, nx, layers)
    assert c.general and not c.has_step
    names = c.get_param_names()
    assert ('scales' in names) == (norm != 'none') and len(names) == len(params_o)
    ps = [p.clone().requires_grad_(True) for p in params_o]
    tg, td, tp, tc = (og.t64(a) for a in (xg, xd, xp, cond))
    loss_o = og.critic_loss(ps, tg, td, tp, tc, tc, tc, 10.0, **kw)
    grads_o = torch.autograd.grad(loss_o, ps)
    flat_o = np.concatenate([g.numpy().ravel() for g in grads_o])
    stats = c.loss_grad(xg, cond, xd, cond, xp, cond, 10.0).cpu().numpy()
    np.testing.assert_allclose(stats[3], float(loss_o), rtol=2e-5, atol=1e-5)
    got = c.grads.cpu().numpy()
    kinked = nonlinearity == 'elu'
    _assert_close(got, flat_o, 1e-3, 2e-5 * np.abs(flat_o).max(), kinked)
    off = 0
    for (kind, shape), g in zip(c.param_shapes(), grads_o):          # ... and tensor by tensor (small tensors hide in a flat comparison)
        n = int(np.prod(shape))
        w = g.numpy().ravel()
        _assert_close(got[off:off + n], w, 2e-3, 1e-4 * max(np.abs(w).max(), 1e-30), kinked, err_msg=kind,
                      loose=1e-3 * np.abs(flat_o).max())
        off += n
    np.testing.assert_allclose(c.forward(xg, cond).cpu().numpy(), og.critic_forward(params_o, tg, tc, **kw)[:, 0].numpy(),
                               rtol=1e-4, atol=1e-5)
    x = tg.clone().requires_grad_(True)
    want = torch.autograd.grad(-og.critic_forward(params_o, x, tc, **kw).mean(), x)[0].numpy()
    gx, dmean = c.input_grad(xg, cond, -1.0 / batch)
    np.testing.assert_allclose(gx.cpu().numpy(), want, rtol=1e-3, atol=2e-5 * np.abs(want).max())
    acc = c.accuracy(xg, cond, xd, cond)
    assert abs(acc - (stats[0] - stats[1])) < 1e-5


@pytest.mark.parametrize('nonlinearity,use_scale', [('leaky_rectify', 'auto'), ('linear', 'auto'), ('rectify', True), ('tanh', False)])
def test_scale_layer_rule_and_option(nonlinearity, use_scale):
    """simple_discriminator.py:57-60: `use_scale='auto'` = every nonlinearity but rectify; the layer option overrides it either
    way (`net_options={'layer': {'use_scale': ...}}`, the reference's `options`).  Piecewise-linear nonlinearities with a
    scale, rectify with a forced scale, tanh with the scale switched off -- all against fp64 autograd."""
    opts = None if use_scale == 'auto' else {'layer': {'use_scale': use_scale}}
    c, params_o, xg, xd, xp, cond, kw = _general_case(nonlinearity, ['none', 'layer'], 40, 6, [24, 16], net_options=opts, use_scale=use_scale)
    want_scale = (nonlinearity != 'rectify') if use_scale == 'auto' else use_scale
    assert ('scales' in c.get_param_names()) == want_scale
    ps = [p.clone().requires_grad_(True) for p in params_o]
    tg, td, tp, tc = (og.t64(a) for a in (xg, xd, xp, cond))
    loss_o = og.critic_loss(ps, tg, td, tp, tc, tc, tc, 10.0, **kw)
    flat_o = np.concatenate([g.numpy().ravel() for g in torch.autograd.grad(loss_o, ps)])
    stats = c.loss_grad(xg, cond, xd, cond, xp, cond, 10.0).cpu().numpy()
    np.testing.assert_allclose(stats[3], float(loss_o), rtol=2e-5, atol=1e-5)
    np.testing.assert_allclose(c.grads.cpu().numpy(), flat_o, rtol=1e-3, atol=2e-5 * np.abs(flat_o).max())


def test_smooth_unconditional_critic_and_bf16():
    """The general path without condition columns (UnConditionalDiscriminator) and with bf16 GEMM operands (bf16 tolerance)."""
    c, params_o, xg, xd, xp, cond, kw = _general_case('tanh', 'layer', 64, 8, [32, 32], conditional=False)
    ps = [p.clone().requires_grad_(True) for p in params_o]
    tg, td, tp = (og.t64(a) for a in (xg, xd, xp))
    loss_o = og.critic_loss(ps, tg, td, tp, None, None, None, 10.0, **kw)
    flat_o = np.concatenate([g.numpy().ravel() for g in torch.autograd.grad(loss_o, ps)])
    stats = c.loss_grad(xg, None, xd, None, xp, None, 10.0).cpu().numpy()
    np.testing.assert_allclose(stats[3], float(loss_o), rtol=2e-5, atol=1e-5)
    np.testing.assert_allclose(c.grads.cpu().numpy(), flat_o, rtol=1e-3, atol=2e-5 * np.abs(flat_o).max())
    cb, params_o, xg, xd, xp, cond, kw = _general_case('elu', 'layer', 256, 8, [128, 128], precision='bf16')
    ps = [p.clone().requires_grad_(True) for p in params_o]
    tg, td, tp, tc = (og.t64(a) for a in (xg, xd, xp, cond))
    loss_o = og.critic_loss(ps, tg, td, tp, tc, tc, tc, 10.0, **kw)
    flat_o = np.concatenate([g.numpy().ravel() for g in torch.autograd.grad(loss_o, ps)])
    stats = cb.loss_grad(xg, cond, xd, cond, xp, cond, 10.0).cpu().numpy()
    np.testing.assert_allclose(stats[3], float(loss_o), rtol=3e-2, atol=3e-2)
    err = np.abs(cb.grads.cpu().numpy() - flat_o).max() / np.abs(flat_o).max()
    assert err < 5e-2, err


@pytest.mark.parametrize('layers,norm,nonlin,precision', [([128, 128, 128, 128], ['none', 'layer', 'layer', 'layer'], 'rectify', 'fp32'),
                                                          ([512, 512, 512], 'none', 'rectify', 'bf16'),
                                                          ([64, 64], 'none', 'leaky_rectify', 'fp32'), ([], 'none', 'rectify', 'fp32')])
def test_one_call_critic_step_equals_the_separate_calls(layers, norm, nonlin, precision):
    """`Critic.step` (`ssn_critic_step_run`: penalty points, loss + gradient, optimizer step, accuracy of the updated critic,
    per-tensor sums of squares and the step's scalar record in one library call) launches the kernels of the separate calls in
    the same order: parameters, optimizer state, gradients and every scalar agree bit for bit over several steps."""
    from tc_gan_amd.critic import Critic, Updater
    rs = np.random.RandomState(7)
    n, nx = 96, 8
    kw = dict(seed=3, precision=precision, normalization=norm, nonlinearity=nonlin)
    a, b = Critic(nx, layers, **kw), Critic(nx, layers, **kw)
    ua, ub = Updater(0.01, 'rmsprop', reg_l2_decay=1e-3), Updater(0.01, 'rmsprop', reg_l2_decay=1e-3)
    for it in range(3):
        xg = torch.as_tensor(rs.rand(n, nx) * 5, device='cuda', dtype=torch.float32)
        xd = torch.as_tensor(rs.rand(n, nx) * 5, device='cuda', dtype=torch.float32)
        cond = torch.as_tensor(np.stack([np.full(n, 20.), rs.rand(n) * 2 - 1, rs.randint(0, 2, n)], axis=1), device='cuda', dtype=torch.float32)
        eps = torch.as_tensor(rs.rand(n, 1), device='cuda', dtype=torch.float32)
        pens = torch.as_tensor(rs.rand(2), device='cuda', dtype=torch.float64)
        xp_a = a.interpolate(eps, xd, xg)
        stats = a.loss_grad(xg, cond, xd, cond, xp_a, cond, 10.0)
        loss_a = stats[3:4].clone()
        ua(a.params, a.grads)
        acc_a = a.accuracy_device(xg, cond, xd, cond)
        want = torch.cat([pens.to(torch.float32), loss_a, acc_a, a.param_sqnorms_device()])
        xp_b, tail = b.step(ub, xg, xd, cond, eps, 10.0, pens64=pens)
        assert torch.equal(xp_a, xp_b) and torch.equal(tail, want), (it, tail, want)
        assert torch.equal(a.params, b.params) and torch.equal(a.grads, b.grads) and ua.step == ub.step
        assert all(torch.equal(x, y) for x, y in zip(ua._state, ub._state))


def test_critic_bf16_path_close_to_fp64():
    from tc_gan_amd.critic import Critic
    c, params_o, xg, xd, xp, cond = _setup(512, 8, [256, 256], seed=3)
    cb = Critic(8, [256, 256], precision='bf16')
    cb.set_flat(c.get_flat())
    tg, td, tp, tc = (og.t64(a) for a in (xg, xd, xp, cond))
    ps = [p.clone().requires_grad_(True) for p in params_o]
    loss_o = og.critic_loss(ps, tg, td, tp, tc, tc, tc, 10.0)
    flat_o = np.concatenate([g.numpy().ravel() for g in torch.autograd.grad(loss_o, ps)])
    stats = cb.loss_grad(xg, cond, xd, cond, xp, cond, 10.0).cpu().numpy()
    # bf16 operands: 8-bit mantissa -> ~1e-2 relative on O(1) quantities
    np.testing.assert_allclose(stats[3], float(loss_o), rtol=3e-2, atol=3e-2)
    got = cb.grads.cpu().numpy()
    err = np.linalg.norm(got - flat_o) / np.linalg.norm(flat_o)
    assert err < 3e-2, err


def test_pipelined_bf16_gemm_agrees_with_the_general_kernel():
    """`gemm_bf16_pipe_kernel` (16-byte loads three K tiles ahead) takes the bf16 GEMMs with K >= 64 and contiguous extents in
    fours; `SSN_GEMM_PIPE=0` sends everything through the general kernel.  Same tile, same k order: loss, gradients and
    critic outputs agree to fp32 summation noise (split-K slices are cut at multiples of 64 instead of 32) for shapes with
    partial tiles in every index: rows not a multiple of 64, widths 72 / 100 / 132 (K tails of 8 / 36 / 4 beyond a tile of
    64), and the C3 shape with its split-K weight gradients."""
    import os
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np, torch; sys.path.insert(0, %r)\n"
        "from tc_gan_amd.critic import Critic\n"
        "res = {}\n"
        "for tag, batch, layers in (('a', 200, [72, 100, 132]), ('b', 1000, [512, 68]), ('c3', 1024, [512, 512, 512])):\n"
        "    rs = np.random.RandomState(len(layers) * 7 + batch)\n"
        "    c = Critic(8, layers, precision='bf16', seed=5)\n"
        "    xg, xd = (torch.as_tensor(rs.rand(batch, 8) * 5, device='cuda', dtype=torch.float32) for _ in range(2))\n"
        "    xp = 0.25 * xg + 0.75 * xd\n"
        "    cond = torch.as_tensor(np.stack([np.full(batch, 20.), rs.rand(batch) * 2 - 1, rs.randint(0, 2, batch)], 1),\n"
        "                           device='cuda', dtype=torch.float32)\n"
        "    res['stats_' + tag] = c.loss_grad(xg, cond, xd, cond, xp, cond, 10.0).cpu().numpy()\n"
        "    res['grads_' + tag] = c.grads.cpu().numpy()\n"
        "    res['out_' + tag] = c.forward(xg, cond).cpu().numpy()\n"
        "np.savez(sys.argv[1], **res)\n" % root)
    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        for form in ('0', '1'):
            path = os.path.join(tmp, 'pipe%s.npz' % form)
            subprocess.run([sys.executable, '-c', code, path], check=True, env=dict(os.environ, SSN_GEMM_PIPE=form), timeout=300)
            res[form] = dict(np.load(path))
    for key, want in res['0'].items():
        assert np.isfinite(want).all() and np.abs(want).max() > 0, key
        np.testing.assert_allclose(res['1'][key], want, rtol=1e-5, atol=2e-6 * np.abs(want).max(), err_msg=key)


def test_rows_path_matches_layer_path():
    """Wide plain critics on bf16 operands run the row-local part of an update as ONE launch (`critic_rows_kernel`,
    ssn_critic_rows.hip: a workgroup walks its 32 rows through forward, backward chain, penalty head and second chain);
    `SSN_CRITIC_ROWS=0` keeps the layer-by-layer chain of GEMM launches.  Same instruction, same operand rounding, same k
    order, same orders of the cross-row sums: statistics, D values and EVERY gradient are equal bit for bit -- at the C3 shape,
    with a ragged last row block, with odd tile counts (widths 96 / 160 / 32), with a leaky nonlinearity, without
    conditions, through the one-call step (parameters after the optimizer, record tail, penalty points), and for the other
    users of the kernel: `forward`, the generator side's `input_grad`, the accuracy of two stacked batches."""
    import os
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np, torch; sys.path.insert(0, %r)\n"
        "from tc_gan_amd.critic import Critic, Updater\n"
        "res = {}\n"
        "cases = (('c3', 1024, [512, 512, 512], 'rectify', True), ('ragged', 1000, [512, 64], 'rectify', True),\n"
        "         ('odd', 77, [96, 160, 32], 'leaky_rectify', True), ('one', 33, [32], 'very_leaky_rectify', True),\n"
        "         ('nocond', 300, [256, 128], 'rectify', False))\n"
        "for tag, batch, layers, nl, conditional in cases:\n"
        "    rs = np.random.RandomState(len(layers) * 7 + batch)\n"
        "    c = Critic(8, layers, precision='bf16', seed=5, nonlinearity=nl, conditional=conditional)\n"
        "    xg, xd = (torch.as_tensor(rs.rand(batch, 8) * 5, device='cuda', dtype=torch.float32) for _ in range(2))\n"
        "    xp = 0.25 * xg + 0.75 * xd\n"
        "    cond = torch.as_tensor(np.stack([np.full(batch, 20.), rs.rand(batch) * 2 - 1, rs.randint(0, 2, batch)], 1),\n"
        "                           device='cuda', dtype=torch.float32) if conditional else None\n"
        "    res['stats_' + tag] = c.loss_grad(xg, cond, xd, cond, xp, cond, 10.0).cpu().numpy()\n"
        "    res['grads_' + tag] = c.grads.cpu().numpy()\n"
        "    res['dvals_' + tag] = c._dvals.cpu().numpy()\n"
        "    half = batch // 2\n"
        "    res['stats2_' + tag] = c.loss_grad(xg[:half], None if cond is None else cond[:half], xd, cond, xp[:half + 3],\n"
        "                                        None if cond is None else cond[:half + 3], 3.0).cpu().numpy()\n"
        "    res['grads2_' + tag] = c.grads.cpu().numpy()\n"
        "    res['fwd_' + tag] = c.forward(xd, cond).cpu().numpy()\n"
        "    gx, dmean = c.input_grad(xg, cond, scale=-1.0 / batch)\n"
        "    res['gx_' + tag] = gx.cpu().numpy(); res['gxmean_' + tag] = dmean.cpu().numpy().reshape(1)\n"
        "    res['acc_' + tag] = c.accuracy_device(xg[:half], None if cond is None else cond[:half], xd, cond).cpu().numpy()\n"
        "    upd = Updater(learning_rate=1e-3, update_name='adam-wgan')\n"
        "    eps = torch.as_tensor(rs.rand(batch), device='cuda', dtype=torch.float32)\n"
        "    for it in range(2):\n"
        "        xp2, tail = c.step(upd, xg, xd, cond, eps, 10.0)\n"
        "    res['step_xp_' + tag] = xp2.cpu().numpy(); res['step_tail_' + tag] = tail.cpu().numpy()\n"
        "    res['step_params_' + tag] = c.params.cpu().numpy(); res['step_stats_' + tag] = c.stats.cpu().numpy()\n"
        "np.savez(sys.argv[1], **res)\n" % root)
    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        for form in ('0', '1'):
            path = os.path.join(tmp, 'rows%s.npz' % form)
            subprocess.run([sys.executable, '-c', code, path], check=True, env=dict(os.environ, SSN_CRITIC_ROWS=form), timeout=300)
            res[form] = dict(np.load(path))
    for key, want in res['0'].items():
        assert np.isfinite(want).all() and np.abs(want).max() > 0, key
        np.testing.assert_array_equal(res['1'][key], want, err_msg=key)


def test_generator_side_input_gradient():
    c, params_o, xg, xd, xp, cond = _setup(96, 8, [64, 64], seed=11)
    x = og.t64(xg).clone().requires_grad_(True)
    loss = -og.critic_forward(params_o, x, og.t64(cond)).mean()
    gx_o, = torch.autograd.grad(loss, x)
    gx, dmean = c.input_grad(xg, cond, scale=-1.0 / 96)
    np.testing.assert_allclose(gx.cpu().numpy(), gx_o.numpy(), rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(float(dmean), -float(loss), rtol=1e-4, atol=1e-6)


def test_hide_cell_type_zeroes_third_condition():
    from tc_gan_amd.critic import Critic
    c = Critic(4, [16], seed=0, precision='fp32', hide_cell_type=True)
    rs = np.random.RandomState(0)
    x = rs.rand(10, 4)
    cond = np.stack([np.full(10, 20.), rs.rand(10), np.zeros(10)], axis=1)
    cond1 = cond.copy(); cond1[:, 2] = 1
    np.testing.assert_array_equal(c.forward(x, cond).cpu().numpy(), c.forward(x, cond1).cpu().numpy())


@pytest.mark.parametrize('name,cfg', [('adam-wgan', {}), ('adam', {}), ('rmsprop', {}), ('sgd', {})])
def test_optimizer_steps_vs_oracle(name, cfg):
    from tc_gan_amd.critic import Updater
    rs = np.random.RandomState(0)
    p = rs.randn(1000).astype(np.float32)
    up = Updater(learning_rate=0.01, update_name=name, reg_l2_penalty=1e-3, reg_l1_penalty=2e-3,
                 reg_l2_decay=1e-2, reg_l1_decay=3e-3)
    pd = torch.tensor(p, device='cuda')
    po = p.astype(np.float64)
    state = {}
    for it in range(5):
        g = rs.randn(1000).astype(np.float32)
        up(pd, torch.tensor(g, device='cuda'), clip=(-2.5, 2.5))
        go = g.astype(np.float64) + 2 * 1e-3 * po + 2e-3 * np.sign(po)
        if name == 'adam-wgan':
            pn = og.adam_step(po, go, state, 0.01, beta1=0.5, beta2=0.9)
        elif name == 'adam':
            pn = og.adam_step(po, go, state, 0.01)
        elif name == 'rmsprop':
            pn = og.rmsprop_step(po, go, state, 0.01)
        else:
            pn = og.sgd_step(po, go, state, 0.01)
        pn = pn - 0.01 * 1e-2 * po - 0.01 * 3e-3 * np.sign(po)       # wgan.py:158-163 (decay on the old value)
        po = np.clip(pn, -2.5, 2.5)
        np.testing.assert_allclose(pd.cpu().numpy(), po, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize('batch,nx,layers,norm', [
    (9, 4, [12], 'layer'), (64, 8, [32, 32], 'layer'), (130, 8, [128, 128, 128, 128], ['none', 'layer', 'layer', 'layer']),
    (48, 8, [24, 24], ['layer', 'none']), (40, 6, [16, 16], 'none')])
def test_layer_normalised_critic_loss_gradient_and_input_gradient(batch, nx, layers, norm):
    """Layer-normalised critic (simple_discriminator.py:6-75; the paper's 4x128 critic normalises layers 2-4):
    loss, parameter gradient INCLUDING the WGAN-GP double backward through the normalisation, and the
    generator-side input gradient, against torch autograd on the fp64 restatement."""
    from tc_gan_amd.critic import Critic
    rs = np.random.RandomState(batch + len(layers))
    c = Critic(nx, layers, seed=batch, precision='fp32', normalization=norm)
    if norm != 'none':
        c.layer_norm = True          # also drive all-plain nets through the general path
    params_o = [og.t64(p) for p in c.get_param_values()]
    xg, xd = rs.rand(batch, nx) * 5, rs.rand(batch, nx) * 5
    eps = rs.rand(batch, 1)
    xp = eps * xd + (1 - eps) * xg
    cond = np.stack([np.full(batch, 20.), rs.rand(batch) * 2 - 1, rs.randint(0, 2, batch)], axis=1)
    tg, td, tp, tc = (og.t64(a) for a in (xg, xd, xp, cond))
    ps = [p.clone().requires_grad_(True) for p in params_o]
    loss_o = og.critic_loss(ps, tg, td, tp, tc, tc, tc, 10.0, normalization=norm)
    flat_o = np.concatenate([g.numpy().ravel() for g in torch.autograd.grad(loss_o, ps)])
    stats = c.loss_grad(xg, cond, xd, cond, xp, cond, 10.0).cpu().numpy()
    np.testing.assert_allclose(stats[3], float(loss_o.detach()), rtol=5e-5, atol=5e-5)
    got = c.grads.cpu().numpy()
    np.testing.assert_allclose(got, flat_o, rtol=2e-3, atol=1e-4 * np.abs(flat_o).max())
    np.testing.assert_allclose(c.forward(xg, cond).cpu().numpy(),
                               og.critic_forward(params_o, tg, tc, normalization=norm)[:, 0].numpy(), rtol=2e-4, atol=2e-5)
    x = tg.clone().requires_grad_(True)
    gx_o, = torch.autograd.grad(-og.critic_forward(params_o, x, tc, normalization=norm).mean(), x)
    gx, _ = c.input_grad(xg, cond, scale=-1.0 / batch)
    np.testing.assert_allclose(gx.cpu().numpy(), gx_o.numpy(), rtol=2e-3, atol=1e-6)


def test_bias_gradients_that_cancel_are_exactly_zero():
    """Units active on every row get +w/B from each generated row and -w/B from each data row: the reference's
    sum is exactly zero.  fp32 partial sums of such terms round (3c, 5c, ... need extra mantissa bits) and leave
    ~1e-8 of noise, which Adam (eps 1e-8) turns into full-size parameter steps; the column sums accumulate in fp64."""
    batch, nx, layers = 24, 8, [16, 16]
    c, params_o, xg, xd, xp, cond = _setup(batch, nx, layers, seed=5)
    flat = c.get_flat().copy()
    off = 0
    bias_slices = []
    dims = [nx + 3] + layers
    for nin, nout in zip(dims[:-1], dims[1:]):
        off += nin * nout
        flat[off:off + nout] = 1000.0                     # every unit active on every row
        bias_slices.append(slice(off, off + nout))
        off += nout
    c.set_flat(flat)
    c.loss_grad(xg, cond, xd, cond, xp, cond, 10.0)
    got = c.grads.cpu().numpy()
    for sl in bias_slices:
        np.testing.assert_array_equal(got[sl], 0.0)


# ---------------------------------------------------------------------------------------------------------------
# The reference's own known answers for this layer (the only reference-held pins of a14 / a11), restated on the
# device kernels.  networks/tests/test_updater.py:24-59 and networks/tests/test_layer_normalization.py:9-65.
# ---------------------------------------------------------------------------------------------------------------
def _update_single_param(updater, before):
    """test_updater.py:9-21: loss = param.mean() * 0, so the gradient is exactly zero and only the regularisers act."""
    p = torch.tensor(np.asarray(before, dtype='float32'), device='cuda')
    g = torch.zeros_like(p)
    updater(p, g)
    return p.cpu().numpy()


def test_reference_updater_known_answers():
    from tc_gan_amd.critic import Updater
    before = np.array([1, 2, 3])
    # test_l2_decay_no_loss: default update ('adam-wgan'), decoupled decay p -= lr * decay * p
    after = _update_single_param(Updater(learning_rate=1, reg_l2_decay=0.2), before)
    np.testing.assert_allclose(after, before * (1 - 0.2), rtol=1e-6)
    # test_l1_decay_no_loss
    after = _update_single_param(Updater(learning_rate=1, reg_l1_decay=0.5), np.array([1, -1]))
    np.testing.assert_allclose(after, [0.5, -0.5], rtol=1e-6)
    # test_l2_penalty_no_loss: sgd on loss + l2 * sum(p^2)  ->  p (1 - 2 l2)
    after = _update_single_param(Updater(learning_rate=1, update_name='sgd', reg_l2_penalty=0.2), before)
    np.testing.assert_allclose(after, before * (1 - 2 * 0.2), rtol=1e-6)
    # test_l1_penalty_no_loss
    after = _update_single_param(Updater(learning_rate=1, update_name='sgd', reg_l1_penalty=0.5), np.array([1, -1]))
    np.testing.assert_allclose(after, [0.5, -0.5], rtol=1e-6)


def _np_norm_layer(x, epsilon=0):
    """test_layer_normalization.py:9-13."""
    kwds = dict(axis=tuple(range(1, len(x.shape))), keepdims=True)
    mean = x.mean(**kwds)
    std = np.sqrt(x.var(**kwds) + epsilon)
    return (x - mean) / std


def _hidden_activations(critic, x):
    """Hidden layer 1 of a one-layer critic, column by column: D = h_1 . w_out with w_out = e_k.  The critic's input is
    [x[:, :-3], contrast, |norm_probe|, cell_type]: the last three columns of `x` go in as the condition."""
    flat = critic.get_flat()
    nout = critic.dims[1]
    cols = []
    for k in range(nout):
        flat[-nout:] = 0.0
        flat[-nout + k] = 1.0
        critic.set_flat(flat)
        cols.append(critic.forward(x[:, :-3], x[:, -3:]).cpu().numpy())
    return np.stack(cols, axis=1)


@pytest.mark.parametrize('batchsize,in_dim,out_dim', [(2, 3 + 3, 4), (10, 30, 20), (5, 11, 128)])
def test_reference_layer_normalized_dense_layer_known_answer(batchsize, in_dim, out_dim):
    """test_layer_normalized_dense_layer (test_layer_normalization.py:41-65): relu(norm(x W) + b) against numpy, with the
    layer's own epsilon (LayerNormLayer inherits BatchNormLayer's 1e-4; the first reference test compares with
    np_norm_layer(x, l1.epsilon), the second sets it to 0 -- both forms are checked, the second to the accuracy that
    eps = 1e-4 allows)."""
    from tc_gan_amd.critic import Critic
    rs = np.random.RandomState(0)
    x = rs.randn(batchsize, in_dim)
    x[:, -2] = np.abs(x[:, -2])                        # the |norm_probe| column enters through abs()
    W = rs.randn(in_dim, out_dim)
    b = rs.randn(out_dim)
    c = Critic(in_dim - 3, [out_dim], precision='fp32', normalization='layer')
    c.set_flat(np.concatenate([W.ravel(), b, np.zeros(out_dim)]))
    actual = _hidden_activations(c, x.astype('float32'))
    a = np.tensordot(x.astype('float32').astype('float64'), W.astype('float32').astype('float64'), axes=1)
    desired = np.maximum(_np_norm_layer(a, 1e-4) + b.astype('float32'), 0)
    assert (desired > 0).any()
    np.testing.assert_allclose(actual, desired, rtol=2e-5, atol=2e-5)
    desired0 = np.maximum(_np_norm_layer(a) + b, 0)    # epsilon = 0 form: differs by O(eps / var)
    np.testing.assert_allclose(actual, desired0, rtol=0, atol=2e-4 * np.abs(desired0).max() / min(1.0, a.var(axis=1).min()))


def test_reference_layer_norm_layer_known_answer():
    """test_layer_norm_layer (test_layer_normalization.py:16-38): the bare LayerNorm output (x - mean)/sqrt(var + eps).
    With W = identity the dense layer is transparent; +b = 0; relu splits into the positive and (by negating the
    input) the negative part."""
    from tc_gan_amd.critic import Critic
    rs = np.random.RandomState(0)
    n = 12
    x = rs.randn(6, n).astype('float32')
    x[:, -2] = np.abs(x[:, -2])
    c = Critic(n - 3, [n], precision='fp32', normalization='layer')
    c.set_flat(np.concatenate([np.eye(n).ravel(), np.zeros(n), np.zeros(n)]))
    pos = _hidden_activations(c, x)
    c.set_flat(np.concatenate([-np.eye(n).ravel(), np.zeros(n), np.zeros(n)]))
    neg = _hidden_activations(c, x)
    np.testing.assert_allclose(pos - neg, _np_norm_layer(x.astype('float64'), 1e-4), rtol=2e-5, atol=2e-6)


def test_layer_by_layer_path_still_covers_small_widths():
    """Critics up to 128 wide (and up to 2048 stacked rows) take the fused row-block kernels (ssn_critic_fused.hip); the
    layer-by-layer MFMA chain keeps serving wider or larger ones.  Rerun this file's parity tests with the fused path
    switched off (SSN_CRITIC_FUSED=0) so that both implementations are checked against the oracle at the same shapes."""
    if os.environ.get('SSN_CRITIC_FUSED') == '0':
        pytest.skip('already running with the fused path disabled')
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-q', '-x', '-m', 'gpu', '-k',
                        'not layer_by_layer'], env=dict(os.environ, SSN_CRITIC_FUSED='0'), capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_step_helpers_segment_sqnorms_and_interpolate():
    """`ssn_segment_sqnorms2_f32` (per-tensor critic statistics, recorders.py:275-311) and `ssn_interpolate_f32`
    (gradient-penalty points, cwgan.py:476-481) against their numpy definitions."""
    from tc_gan_amd.critic import Critic
    crit = Critic(8, [16, 32, 8], normalization='layer', seed=3)
    got = crit.param_sqnorms_device().cpu().numpy()
    want = [float((np.asarray(v, dtype='float64') ** 2).sum()) for v in crit.get_param_values()]
    np.testing.assert_allclose(got, want, rtol=1e-6)
    assert len(got) == len(crit.get_param_names())
    # the full-size critic of BASELINE config 3 (0.53 M parameters, tensors from 1 to 262144 elements; bias and single-row
    # tensors leave most chunks empty): same values, and the same bits on every call (fixed summation order, no atomics)
    big = Critic(8, [512, 512, 512], seed=5)
    g1 = big.param_sqnorms_device().cpu().numpy()
    want = [float((np.asarray(v, dtype='float64') ** 2).sum()) for v in big.get_param_values()]
    np.testing.assert_allclose(g1, want, rtol=1e-6)
    for _ in range(3):
        np.testing.assert_array_equal(big.param_sqnorms_device().cpu().numpy(), g1)
    # the symbol's older signature (no scratch argument; include/ssnode_mi355x.h): same bits
    from tc_gan_amd import clib
    old = torch.empty(len(g1), device='cuda', dtype=torch.float32)
    clib.check(clib.libssnode.ssn_segment_sqnorms_f32(big.params.data_ptr(), big._seg_bounds.data_ptr(), len(g1), old.data_ptr(),
                                                      clib.stream_ptr()), 'ssn_segment_sqnorms_f32')
    np.testing.assert_array_equal(old.cpu().numpy(), g1)
    rs = np.random.RandomState(0)
    eps, xd, xg = rs.rand(37, 1).astype('float32'), rs.randn(37, 8).astype('float32'), rs.randn(37, 8).astype('float32')
    xp = crit.interpolate(torch.as_tensor(eps).cuda(), torch.as_tensor(xd).cuda(), torch.as_tensor(xg).cuda()).cpu().numpy()
    np.testing.assert_array_equal(xp, eps * xd + (np.float32(1) - eps) * xg)
