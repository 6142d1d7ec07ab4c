"""GPU parity of the critic (forward, WGAN-GP loss and gradient, input gradient) and of the optimizer
kernels against oracle/gan_torch.py.  fp32-MFMA path: tight; bf16-MFMA path: bf16 tolerance."""
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
