"""WGAN fit of the feed-forward tuning-curve model on the GPU -- the training part of the reference's
``FF_lalazar_model.py`` (``make_WGAN_funcs`` 398-454, ``train_wgan`` 223-330), on top of the generator kernels of
`ff_model`.

The script's critic is ``SD.make_net(INSHAPE, "WGAN")`` with no hidden layers: one bias-free linear unit on the
27 stimulus responses, D(x) = w . x.  Its loss (FF_lalazar_model.py:419-437):

    mean D(x_g) - mean D(x_d) + lam * mean_b (|| d D(log(1 + x_p)) / d x_p ||_2 - 1)^2 + plam * sum(w^2)

(the log(1 + x) is applied on the PENALTY input only), lam = plam = 1, Adam(lr .01, beta1 .5, beta2 .9) for both
players; the generator minimises -mean D(G(z)) over the five log-space parameters (PARAM, line 103).  With a linear
critic every gradient is closed-form: a handful of 27-vectors (device tensor algebra); the heavy part -- the
generator forward over box^3 grid points and its backward -- are the HIP kernels of csrc/ssn_ff.hip.

Host RNG consumption per critic step follows the script (two F_gen() draws -- the second is unused there too --,
the data choice, the interpolation coefficients) so that a seeded run visits the same minibatches.
"""
import os

import numpy as np
import torch

from . import clib, ff_model
from .critic import Updater


def read_dat(path):
    """FF_lalazar_model.py:38-50: comma-separated floats, one row per line."""
    return np.array([[float(t) for t in line.strip().split(',')] for line in open(path) if line.strip()])


class FFWGAN(object):
    def __init__(self, box_width, curves, nsam=20, nhid=1, seed=1, lam=1.0, plam=1.0, learning_rate=0.01,
                 start_params=None):
        clib.require_gpu()
        self.box_width, self.nsam, self.nhid = int(box_width), int(nsam), int(nhid)
        self.rng = seed if isinstance(seed, np.random.RandomState) else np.random.RandomState(seed)
        self.curves = np.asarray(curves, dtype='float64')
        self.stim = ff_model.default_stimuli()
        self.ni = len(self.stim)
        assert self.curves.shape[1] == self.ni
        self.lam, self.plam = lam, plam
        self.params = dict(ff_model.START_PARAMS, **(start_params or {}))
        # make_mask (FF_lalazar_model.py:188-195): NOBS = 1 observed unit out of NHID
        self.observed = int(self.rng.choice(np.arange(self.nhid), 1)[0])
        a = np.sqrt(6.0 / (self.ni + 1))                       # Lasagne GlorotUniform, DenseLayer(27 -> 1, b=None)
        self.w = torch.as_tensor(self.rng.uniform(-a, a, self.ni), device='cuda', dtype=torch.float32)
        self.d_updater = Updater(learning_rate, 'adam-wgan')
        self.g_updater = Updater(learning_rate, 'adam-wgan')
        self._gp = torch.zeros(len(ff_model.PARAM_NAMES), device='cuda', dtype=torch.float32)

    # -- generator ------------------------------------------------------------------------------------
    def draw(self):
        return ff_model.generate_samples(self.rng, self.nsam, self.box_width, self.nhid)

    def generate(self, sample, keep=False):
        con, strn, wid, ths = sample
        # (the draw is the script's -- dense arrays, its RNG consumption -- the kernels take the unit's box^3 / 100 connections
        # as lists: `ssn_ff_forward_sparse_f32`, a third of the dense entry point's bytes)
        idx, val = ff_model.sparsify(con, strn)
        res = ff_model.ff_forward_sparse(self.params, wid, idx, val, ths, self.stim, self.box_width, keep=keep)
        out, saved = res if keep else (res, None)
        return out[:, :, self.observed], out, saved          # (nsam, ni) observed responses

    # -- critic (closed forms) ------------------------------------------------------------------------
    def critic_loss_grad(self, xd, xg, xp):
        """(Wasserstein distance estimate, full loss, d loss / d w) for minibatches (nsam, ni)."""
        w = self.w.to(torch.float64)
        xd, xg, xp = (t.to(torch.float64) for t in (xd, xg, xp))
        wdist = (xg * w[None, :]).sum(dim=1).mean() - (xd * w[None, :]).sum(dim=1).mean()   # 27-term dots: elementwise
        a = 1.0 / (1.0 + xp)                                  # d log(1 + x) / d x
        nrm = torch.sqrt(((w[None, :] * a) ** 2).sum(dim=1))
        pen = ((nrm - 1.0) ** 2).mean()
        dpen = (2.0 * (nrm - 1.0) / nrm)[:, None] * (w[None, :] * a * a)
        grad = xg.mean(dim=0) - xd.mean(dim=0) + self.lam * dpen.mean(dim=0) + 2.0 * self.plam * w
        loss = wdist + self.lam * pen + self.plam * (w * w).sum()
        return float(wdist), float(loss), grad.to(torch.float32)

    def critic_step(self):
        """One iteration of the inner loop of train_wgan (FF_lalazar_model.py:262-285)."""
        ss = self.draw()
        self.draw()                                           # "DD = F_gen()": drawn and never used by the script
        idx = self.rng.choice(np.arange(len(self.curves)), self.nsam)
        xd = torch.as_tensor(self.curves[idx], device='cuda', dtype=torch.float32)
        xg, _, _ = self.generate(ss)
        ee = torch.as_tensor(self.rng.rand(self.nsam, 1), device='cuda', dtype=torch.float32)
        xp = ee * xd + (1.0 - ee) * xg
        wdist, loss, grad = self.critic_loss_grad(xd, xg, xp)
        self.d_updater(self.w, grad.contiguous())
        return wdist

    def generator_step(self):
        """FF_lalazar_model.py:291-292: loss = -mean D(G(z)); Adam on (RF_low, RF_del, THR, THR_del, Js)."""
        ss = self.draw()
        xg, out, saved = self.generate(ss, keep=True)
        loss = -float((xg.to(torch.float64) * self.w.to(torch.float64)[None, :]).sum(dim=1).mean())
        g_out = torch.zeros_like(out)
        g_out[:, :, self.observed] = -self.w[None, :] / self.nsam
        g = ff_model.ff_backward(self.params, saved, out, g_out)
        vals = torch.as_tensor([self.params[n] for n in ff_model.PARAM_NAMES], dtype=torch.float32)
        self._gp.copy_(vals)
        grads = torch.as_tensor([g[n] for n in ff_model.PARAM_NAMES], device='cuda', dtype=torch.float32)
        self.g_updater(self._gp, grads)
        for n, v in zip(ff_model.PARAM_NAMES, self._gp.cpu().numpy()):
            self.params[n] = float(v)
        return loss, ss

    # -- loop with the script's log files -----------------------------------------------------------
    def train(self, niter, outdir='.', n_critic=5, n_critic_first=500, tag=None, save_interval=1000):
        tag = tag or 'wgan_FF_{0}_{0}'.format(self.box_width)
        os.makedirs(os.path.join(outdir, 'FF_logs'), exist_ok=True)
        os.makedirs(os.path.join(outdir, 'disc_params'), exist_ok=True)
        log = os.path.join(outdir, 'FF_logs', 'FF_log_' + tag + '.csv')
        losslog = os.path.join(outdir, 'FF_logs', 'FF_losslog_' + tag + '.csv')
        tcfile = os.path.join(outdir, 'tuning_curves' + tag + '.csv')
        with open(log, 'w') as f:
            f.write('RF\tRFd\tJ\tth\tth_d\n')
        with open(losslog, 'w') as f:
            f.write('gloss,dloss\n')
        with open(tcfile, 'w') as f:
            f.write('tuning curves for ' + tag)
        gloss = dloss = 10.0
        for k in range(niter):
            for _ in range(n_critic_first if k == 0 else n_critic):
                dloss = self.critic_step()
                with open(losslog, 'a') as f:
                    f.write('{},{}\n'.format(gloss, dloss))
            gloss, ss = self.generator_step()
            with open(losslog, 'a') as f:
                f.write('{},{}\n'.format(gloss, dloss))
            if k % save_interval == 0:
                np.save(os.path.join(outdir, 'disc_params', 'D_par_{}_'.format(k) + tag), [self.w.cpu().numpy()])
            p = self.params
            with open(log, 'a') as f:
                f.write('\t'.join(str(np.round(p[n], 10)) for n in ('RF_low', 'RF_del', 'Js', 'THR', 'THR_del', 'As')) + '\n')
            curv = self.generate(ss)[0].cpu().numpy().reshape(self.nsam, self.ni)
            with open(tcfile, 'a') as f:
                for c in curv:
                    f.write(','.join(str(v) for v in c) + '\n')
        return gloss, dloss


def load_curves(data_dir, rng):
    """FF_lalazar_model.py:75-86: both tuning-curve files, shuffled, first half for training."""
    allcurves = np.concatenate([read_dat(os.path.join(data_dir, 'TuningCurvesFull_Pronation.dat')),
                                read_dat(os.path.join(data_dir, 'TuningCurvesFull_Supination.dat'))], axis=0)
    rng.shuffle(allcurves)
    half = allcurves.shape[0] // 2
    return allcurves[:half], allcurves[half:]


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description=__doc__.split('\n\n')[0])
    ap.add_argument('box_width', type=int, help='grid points along each dimension (XPOINTS of the reference script)')
    ap.add_argument('--data', default='lalazar_data', help='directory holding TuningCurvesFull_{Pronation,Supination}.dat')
    ap.add_argument('--niter', type=int, default=100000)
    ap.add_argument('--nsam', type=int, default=20)
    ap.add_argument('--outdir', default='.')
    ap.add_argument('--seed', type=int, default=1)
    ns = ap.parse_args(argv)
    rng = np.random.RandomState(ns.seed)
    curves, test = load_curves(ns.data, rng)
    np.savetxt(os.path.join(ns.outdir, 'FF_test_curves.csv'), test)
    gan = FFWGAN(ns.box_width, curves, nsam=ns.nsam, seed=rng)
    gan.train(ns.niter, outdir=ns.outdir)


if __name__ == '__main__':
    main()
