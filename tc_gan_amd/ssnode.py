"""SSN fixed-point API on MI355X -- host-side mirror of ``tc_gan/ssnode.py``.

Same names, argument meaning and error behaviour as the reference module
(citations are to /root/reference/tc_gan/ssnode.py), with the Euler solver
running in HIP kernels behind ``tc_gan_amd.clib``:

* `fixed_point` (ssnode.py:159-273) marshals ONE solve through the reference's
  own C entry points ``solve_dynamics_{io_type}_{solver}`` (now GPU, fp64).
* `find_fixed_points` (ssnode.py:332-510) keeps the rejection-sampling
  contract -- the first `num` draws, in draw order, whose solves succeed for
  every stimulus -- but evaluates whole rounds of candidate draws with one
  batched launch (`fixed_points_batch`) instead of a thread pool of
  single-solve ctypes calls.
* `sample_fixed_points` / `sample_tuning_curves` (ssnode.py:561-602) draw ``z``
  with the same ``RandomState`` stream and build W on the device.

No computation happens on the CPU: without the HIP library or a GPU the calls
raise (``OSError`` at import, `GPUUnavailableError` at call).
"""
from __future__ import print_function, division

import collections
import ctypes
import itertools

import numpy as np

from . import clib
from .clib import libssnode, double_ptr

DEFAULT_PARAMS = dict(
    N=102,
    J=np.array([[.0957, .0638], [.1197, .0479]]),
    D=np.array([[.7660, .5106], [.9575, .3830]]),
    S=np.array([[.6667, .2], [1.333, .2]]) / 8,
    bandwidths=[0, 0.0625, 0.125, 0.1875, 0.25, 0.5, 0.75, 1],
    smoothness=0.25 / 8,
    contrast=[20],
    offset=[0],
    io_type='asym_tanh',
    k=0.01,
    n=2.2,
    rate_soft_bound=200, rate_hard_bound=1000,
    tau=(0.01589, 0.002),  # (reference comment: integer ratio tau_E/I is bad)
)


class FixedPointResult(object):
    """ssnode.py:44-56."""

    message = None

    def __init__(self, x, error, steps=None):
        self.x = x
        self.error = error
        self.steps = steps

    @property
    def success(self):
        return self.error == 0

    def to_exception(self):
        return FixedPointError(self.message, self)


class FixedPointError(Exception):
    """ssnode.py:59-63."""

    def __init__(self, message, result):
        super(FixedPointError, self).__init__(message)
        self.result = result


def _message_for(error):
    # ssnode.py:256-270
    if error == 0:
        return "Converged"
    elif error == 1:
        return "SSN Convergence Failed"
    elif error == 2:
        return "Reached to rate_stop_at"
    elif error > 900:
        return "GSL error {}".format(error - 1000)
    return "Unknown error: code={}".format(error)


def make_neu_vec(N, E, I):
    """ssnode.py:84-88."""
    return np.array([E] * N + [I] * N)


def any_to_neu_vec(N, vec):
    vec = np.asarray(vec)
    if len(vec) == 2:
        vec = make_neu_vec(N, *vec)
    return vec


# --------------------------------------------------------------------------
# I/O nonlinearities on arrays (device evaluated)
# --------------------------------------------------------------------------
def _torch():
    import torch
    return torch


def _device_tensor(a, dtype):
    torch = _torch()
    clib.require_gpu()
    if isinstance(a, torch.Tensor):
        return a.to(device='cuda', dtype=dtype).contiguous()
    return torch.as_tensor(np.ascontiguousarray(a)).to(device='cuda', dtype=dtype).contiguous()


def _stream_ptr():
    torch = _torch()
    return clib.stream_ptr()


def _params(io_type, k, n, tau=(1., 1.), dt=1., max_iter=0, atol=0.,
            rate_soft_bound=DEFAULT_PARAMS['rate_soft_bound'],
            rate_hard_bound=DEFAULT_PARAMS['rate_hard_bound']):
    if io_type not in clib.IO_CODES:
        raise ValueError("Unknown I/O type: {}".format(io_type))
    return clib.SolverParams(
        io_type=clib.IO_CODES[io_type], max_iter=int(max_iter), k=float(k), n=float(n),
        tau_E=float(tau[0]), tau_I=float(tau[1]), dt=float(dt), atol=float(atol),
        rate_soft_bound=float(rate_soft_bound), rate_hard_bound=float(rate_hard_bound))


def io_eval(v, io_type, k, n, rate_soft_bound=DEFAULT_PARAMS['rate_soft_bound'],
            rate_hard_bound=DEFAULT_PARAMS['rate_hard_bound']):
    """Evaluate the I/O nonlinearity elementwise on the GPU (numpy in -> numpy
    out, torch CUDA tensor in -> torch CUDA tensor out)."""
    torch = _torch()
    is_tensor = isinstance(v, torch.Tensor)
    if is_tensor:
        dtype = v.dtype if v.dtype in (torch.float32, torch.float64) else torch.float64
    else:
        v = np.asarray(v)
        dtype = torch.float32 if v.dtype == np.float32 else torch.float64
    dv = _device_tensor(v, dtype)
    out = torch.empty_like(dv)
    p = _params(io_type, k, n, rate_soft_bound=rate_soft_bound, rate_hard_bound=rate_hard_bound)
    fn = libssnode.ssn_io_eval_f32 if dtype == torch.float32 else libssnode.ssn_io_eval_f64
    clib.check(fn(dv.data_ptr(), out.data_ptr(), dv.numel(), ctypes.byref(p), _stream_ptr()), 'ssn_io_eval')
    if is_tensor:
        return out
    return out.cpu().numpy().reshape(np.shape(v))


def rate_to_volt(rate, k, n):
    """ssnode.py:125-126 (host scalar arithmetic on parameters only)."""
    return (rate / k)**(1 / n)


def io_alin(v, volt_max, k, n):
    """ssnode.py:129-134; `volt_max` is v0."""
    return io_eval(v, 'asym_linear', k, n, rate_soft_bound=k * volt_max**n)


def io_power(v, k, n):
    """ssnode.py:137-139."""
    return io_eval(v, 'asym_power', k, n)


def io_atanh(v, r0, r1, v0, k, n):
    """ssnode.py:142-149 (v0 must equal rate_to_volt(r0, k, n), as in every reference caller)."""
    return io_eval(v, 'asym_tanh', k, n, rate_soft_bound=r0, rate_hard_bound=r1)


def make_io_fun(k, n,
                rate_soft_bound=DEFAULT_PARAMS['rate_soft_bound'],
                rate_hard_bound=DEFAULT_PARAMS['rate_hard_bound'],
                io_type=DEFAULT_PARAMS['io_type']):
    """ssnode.py:276-292."""
    if io_type not in clib.IO_CODES:
        raise ValueError("Unknown I/O type: {}".format(io_type))

    def io_fun(v):
        return io_eval(v, io_type, k, n, rate_soft_bound=rate_soft_bound, rate_hard_bound=rate_hard_bound)
    return io_fun


# --------------------------------------------------------------------------
# single solve through the reference's C entry points
# --------------------------------------------------------------------------
def solve_dynamics(*args, **kwds):
    """ssnode.py:152-156: the state only; a failed solve is reported on stdout, not raised."""
    outcome = fixed_point(*args, **kwds)
    if outcome.error != 0:
        print(outcome.message)
    return outcome.x


_IO_TYPES = ('asym_linear', 'asym_tanh', 'asym_power')
_UNBOUNDED_IO = ('asym_power', 'asym_linear')      # their only rate bound is the caller's rate_stop_at (ssnode.py:241-242)


def _legacy_entry(io_type, solver):
    """The drop-in C symbol of one (I/O function, solver) pair, looked up by name like the reference does
    (ssnode.py:244-245: 'solve_dynamics_{io_type}_{solver}')."""
    if io_type not in _IO_TYPES:
        raise ValueError("Unknown I/O type: {}".format(io_type))
    if solver not in ('euler'):
        raise ValueError("Unknown solver: {}".format(solver))
    return getattr(libssnode, 'solve_dynamics_' + io_type + '_' + solver)


def _single_solve_buffers(W, ext, r0):
    """fp64, C-contiguous views of one solve's inputs plus the two state buffers the C routine ping-pongs between.
    W and ext are passed through when they already qualify; the initial state is ALWAYS a private copy, because the
    routine writes its result into it and that buffer becomes `FixedPointResult.x` (gotcha 6)."""
    W = np.ascontiguousarray(W, dtype=np.float64)
    if W.ndim != 2 or W.shape[0] != W.shape[1] or W.shape[0] % 2:
        raise AssertionError('W must be a (2N, 2N) matrix, got shape {}'.format(W.shape))
    M = W.shape[0]
    ext = np.ascontiguousarray(ext, dtype=np.float64)
    state = np.zeros(M) if r0 is None else np.array(r0, dtype=np.float64, order='C')
    if ext.shape != (M,) or state.shape != (M,):
        raise AssertionError('ext and r0 must have shape ({},), got {} and {}'.format(M, ext.shape, state.shape))
    return M // 2, W, ext, state, np.empty(M)


def fixed_point(
        W, ext, k, n, r0=None, tau=DEFAULT_PARAMS['tau'],
        max_iter=10000, atol=1e-5, dt=.0008, solver='euler',
        rate_soft_bound=DEFAULT_PARAMS['rate_soft_bound'],
        rate_hard_bound=DEFAULT_PARAMS['rate_hard_bound'],
        rate_stop_at=np.inf,
        io_type='asym_tanh', check=False):
    """
    One fixed-point solve through the reference's own C entry point (ssnode.py:159-273): same parameters, same
    `FixedPointResult` (``x``, ``error``, ``message``, ``success``).  The symbol executes the fp64 HIP kernel.

    error: 0 converged; 1 `max_iter` reached -- or "converged" onto a non-finite state, which the reference also books as
    a failure; 2 a rate reached `rate_stop_at` (power-law / linear I/O only).  ``check=True`` raises `FixedPointError`.
    """
    entry = _legacy_entry(io_type, solver)
    N, W, ext, state, scratch = _single_solve_buffers(W, ext, r0)
    stop_bound = rate_stop_at if io_type in _UNBOUNDED_IO else rate_hard_bound
    clib.require_gpu()
    code = entry(N, W.ctypes.data_as(double_ptr), ext.ctypes.data_as(double_ptr), float(k), float(n),
                 state.ctypes.data_as(double_ptr), scratch.ctypes.data_as(double_ptr),
                 float(tau[0]), float(tau[1]), dt, int(max_iter), atol, rate_soft_bound, stop_bound)
    if code > 900:                      # the library's own failures (no device, launch error) are not solver outcomes
        raise clib.SSNLibraryError('solve_dynamics_{}_{}: status {} ({})'.format(io_type, solver, code, clib.last_error()))
    outcome = FixedPointResult(state, code)
    outcome.message = _message_for(code)
    if code == 0 and not np.all(np.isfinite(state)):
        outcome.error, outcome.message = 1, "Converged to non-finite value"
    if check and outcome.error != 0:
        raise outcome.to_exception()
    return outcome


# --------------------------------------------------------------------------
# batched solve (the GPU hot path)
# --------------------------------------------------------------------------
BatchResult = collections.namedtuple('BatchResult', ['x', 'codes', 'steps', 'x_prev'])


def fixed_points_batch(
        W, exts, k, n, r0=None, tau=DEFAULT_PARAMS['tau'],
        max_iter=10000, atol=1e-5, dt=.0008, solver='euler',
        rate_soft_bound=DEFAULT_PARAMS['rate_soft_bound'],
        rate_hard_bound=DEFAULT_PARAMS['rate_hard_bound'],
        rate_stop_at=np.inf, io_type='asym_tanh',
        dtype='float64', variant=None, return_torch=False, want_prev=False):
    """
    Solve B weight draws x NB stimuli with one launch (additive API).

    Parameters mirror `fixed_point`; additionally

    W : array or CUDA tensor of shape (B, 2N, 2N)
    exts : (NB, 2N), shared by every draw, or (B, NB, 2N)
    r0 : None (zeros), (2N,), or (B, NB, 2N)
    dtype : 'float64' (reference arithmetic) or 'float32' (fast path)
    variant : None (auto), 0 streaming, 1 register-stationary DPP, 2 tile (shape chosen by the library), 3 tile with
        split VGPR/LDS residency, 4 tile with the whole tile in VGPRs, 5 fp32 MFMA kernel (NB >= 4)

    Returns `BatchResult` with ``x`` (B, NB, 2N) newest states, ``codes`` and
    ``steps`` (B, NB) -- codes as the C solver: 0 converged, 1 max_iter,
    2 reached rate_stop_at.
    """
    torch = _torch()
    if io_type not in clib.IO_CODES:
        raise ValueError("Unknown I/O type: {}".format(io_type))
    if solver not in ('euler'):
        raise ValueError("Unknown solver: {}".format(solver))
    tdtype = {'float64': torch.float64, 'float32': torch.float32}[str(np.dtype(dtype))]
    dW = _device_tensor(W, tdtype)
    assert dW.dim() == 3 and dW.shape[1] == dW.shape[2] and dW.shape[1] % 2 == 0
    B, M = int(dW.shape[0]), int(dW.shape[1])
    dE = _device_tensor(exts, tdtype)
    if dE.dim() == 2:
        ext_per_draw, NB = 0, int(dE.shape[0])
    else:
        assert dE.dim() == 3 and dE.shape[0] == B
        ext_per_draw, NB = 1, int(dE.shape[1])
    assert dE.shape[-1] == M
    if r0 is None:
        dR = torch.zeros((B, NB, M), device='cuda', dtype=tdtype)
    else:
        dR = _device_tensor(r0, tdtype)
        dR = dR.expand(B, NB, M).contiguous().clone() if dR.dim() < 3 else dR.clone()
    dP = torch.empty_like(dR) if want_prev else None
    codes = torch.empty((B, NB), device='cuda', dtype=torch.int32)
    steps = torch.empty((B, NB), device='cuda', dtype=torch.int32)
    if io_type in ('asym_power', 'asym_linear'):     # ssnode.py:241-242
        rate_hard_bound = rate_stop_at
    p = _params(io_type, k, n, tau=tau, dt=dt, max_iter=max_iter, atol=atol,
                rate_soft_bound=rate_soft_bound, rate_hard_bound=rate_hard_bound)
    suffix = 'f64' if tdtype == torch.float64 else 'f32'
    args = [dW.data_ptr(), dE.data_ptr(), ext_per_draw, dR.data_ptr(),
            dP.data_ptr() if dP is not None else None, codes.data_ptr(), steps.data_ptr(),
            B, NB, M, ctypes.byref(p), _stream_ptr()]
    if variant is None:
        rc = getattr(libssnode, 'ssn_solve_batch_' + suffix)(*args)
    else:
        rc = getattr(libssnode, 'ssn_solve_batch_{}_variant'.format(suffix))(int(variant), *args)
    clib.check(rc, 'ssn_solve_batch_' + suffix)
    if return_torch:
        return BatchResult(dR, codes, steps, dP)
    torch.cuda.synchronize()
    return BatchResult(dR.cpu().numpy(), codes.cpu().numpy(), steps.cpu().numpy(),
                       dP.cpu().numpy() if dP is not None else None)


FixedPointsInfo = collections.namedtuple('FixedPointsInfo', [
    'solutions', 'counter', 'rejections', 'unused',
])
null_FixedPointsInfo = FixedPointsInfo(None, None, 0, 0)


def _take(n, iterable):
    return list(itertools.islice(iterable, n))


def find_fixed_points(num, Z_W_gen, exts, method='parallel', **common_kwargs):
    """
    Find `num` sets of fixed points using weight matrices from `Z_W_gen`
    (ssnode.py:332-387; same parameters and return values).

    `method` is accepted for compatibility ('parallel' and 'serial' in the
    reference); every method runs the batched GPU solver.  The accepted set is
    the one the reference's serial finder (390-420) and its deterministic
    parallel finder (423-510) both return: the first `num` draws, in the order
    `Z_W_gen` yields them, for which every stimulus converges.  A rejected
    draw is counted under the error code of its first failing stimulus in
    REVERSED stimulus order (largest bandwidth first, ssnode.py:393-394).
    """
    if method not in ('parallel', 'serial', 'batched'):
        raise ValueError('Unknown method: {}'.format(method))
    return find_fixed_points_batched(num, Z_W_gen, exts, **common_kwargs)


def _classify_round(x, codes, steps):
    """Per draw: (ok, error_code_of_first_failure_in_reversed_order)."""
    finite = np.isfinite(x).all(axis=-1)
    err = np.where((codes == 0) & ~finite, 1, codes)   # "Converged to non-finite value" -> 1
    out = []
    for b in range(err.shape[0]):
        bad = np.nonzero(err[b, ::-1])[0]
        out.append((len(bad) == 0, int(err[b, ::-1][bad[0]]) if len(bad) else 0))
    return out, err


def _solve_round(Ws, exts, dtype, kwargs):
    """One round of candidate draws.  Under torch.distributed (world size G > 1, every rank running the same finder
    on the same generator) the draws are dealt round-robin -- rank r solves draws r, r + G, ... -- and the per-draw
    results are exchanged with one all-gather, so that every rank applies the "first `num` successes in submission
    order" rule to the same complete round and returns the same sample."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return fixed_points_batch(Ws, exts, dtype=dtype, **kwargs)
    rank, world = dist.get_rank(), dist.get_world_size()
    mine = np.arange(rank, len(Ws), world)
    local = fixed_points_batch(Ws[mine], exts, dtype=dtype, **kwargs)
    parts = [None] * world
    dist.all_gather_object(parts, (np.asarray(local.x), np.asarray(local.codes), np.asarray(local.steps)))
    x = np.empty((len(Ws),) + parts[0][0].shape[1:], dtype=parts[0][0].dtype)
    codes = np.empty((len(Ws),) + parts[0][1].shape[1:], dtype=parts[0][1].dtype)
    steps = np.empty((len(Ws),) + parts[0][2].shape[1:], dtype=parts[0][2].dtype)
    for r, (px, pc, ps) in enumerate(parts):
        idx = np.arange(r, len(Ws), world)
        x[idx], codes[idx], steps[idx] = px, pc, ps
    return BatchResult(x, codes, steps, None)


def find_fixed_points_batched(num, Z_W_gen, exts, resubmit_threshold=0,
                              deterministic=True, no_pool=False, check=False,
                              dtype='float64', round_size=None,
                              **common_kwargs):
    """Batched rejection sampling; see `find_fixed_points`.  ``no_pool``,
    ``deterministic`` and ``resubmit_threshold`` are accepted and ignored (the
    batched finder is always deterministic)."""
    exts = np.asarray(exts, dtype='double')
    Z_W_gen = iter(Z_W_gen)
    counter = collections.Counter()
    accepted = []          # (Z, [FixedPointResult per stimulus])
    consumed = 0
    used = 0               # draws examined in order until the num-th success
    done = False
    while not done:
        needed = num - len(accepted)
        count = needed if consumed == 0 else max(int(needed * 1.5) + 1, 8)
        if round_size:
            count = max(count, int(round_size)) if consumed else count
        batch = _take(count, Z_W_gen)
        if not batch:
            break
        consumed += len(batch)
        Ws = np.stack([np.asarray(W, dtype='double') for _, W in batch])
        res = _solve_round(Ws, exts, dtype, common_kwargs)
        verdicts, err = _classify_round(res.x, res.codes, res.steps)
        for b, (ok, code) in enumerate(verdicts):
            used += 1
            if ok:
                sols = []
                for s in range(exts.shape[0]):
                    sol = FixedPointResult(np.asarray(res.x[b, s], dtype='double'), 0, int(res.steps[b, s]))
                    sol.message = _message_for(0)
                    sols.append(sol)
                accepted.append((batch[b][0], sols))
                if len(accepted) >= num:
                    done = True
                    break
            else:
                counter[code] += 1
                if check:
                    bad = FixedPointResult(np.asarray(res.x[b], dtype='double'), code)
                    bad.message = _message_for(code)
                    raise bad.to_exception()
    if len(accepted) < num and not accepted:
        raise ValueError('Z_W_gen was exhausted before any fixed point was found')
    zs = np.array([z for z, _ in accepted])
    xs = np.array([[s.x for s in sols] for _, sols in accepted])
    solutions = tuple(sols for _, sols in accepted)
    return zs, xs, FixedPointsInfo(solutions, counter, sum(counter.values()), consumed - used)


# The reference exposes both finders by name; keep them importable.
find_fixed_points_serial = find_fixed_points_batched
find_fixed_points_parallel = find_fixed_points_batched


def _weight_draws(noise_source, N, J, D, S):
    """Endless (z, W) pairs: one ``rand(1, 2N, 2N)`` call per candidate draw from the given RandomState, in draw order
    (the stream contract of ssnode.py:580-586), W from `weight_gen.generate_weight` (the reference goes through a
    compiled Theano expression here, `numeric_w`; equal to 2.6e-11)."""
    from .weight_gen import generate_weight
    M = 2 * N
    while True:
        z = noise_source.rand(1, M, M)[0]
        yield z, generate_weight(N, J, D, S, z)


def _stimulus_rows(N, bandwidths, smoothness, contrast, offset=(0,)):
    from . import stimuli
    return stimuli.input(bandwidths, np.linspace(-0.5, 0.5, N), smoothness, contrast, offset)


def make_solver_params(
        N=DEFAULT_PARAMS['N'],
        J=DEFAULT_PARAMS['J'],
        D=DEFAULT_PARAMS['D'],
        S=DEFAULT_PARAMS['S'],
        io_type=DEFAULT_PARAMS['io_type'],
        seed=65,
        bandwidth=1,
        smoothness=DEFAULT_PARAMS['smoothness'],
        contrast=DEFAULT_PARAMS['contrast'],
        k=DEFAULT_PARAMS['k'],
        n=DEFAULT_PARAMS['n'],
        ):
    """Keyword arguments of one `fixed_point` call on a seeded random network (ssnode.py:524-558); `seed` may be an int
    or a RandomState to draw from."""
    source = seed if hasattr(seed, 'rand') else np.random.RandomState(seed)
    _, W = next(_weight_draws(source, N, J, D, S))
    return dict(W=W, ext=_stimulus_rows(N, [bandwidth], smoothness, contrast)[0], r0=np.zeros(2 * N), k=k, n=n,
                io_type=io_type)


def sample_fixed_points(
        NZ=30, seed=0,
        N=DEFAULT_PARAMS['N'],
        J=DEFAULT_PARAMS['J'],
        D=DEFAULT_PARAMS['D'],
        S=DEFAULT_PARAMS['S'],
        bandwidths=DEFAULT_PARAMS['bandwidths'],
        smoothness=DEFAULT_PARAMS['smoothness'],
        contrast=DEFAULT_PARAMS['contrast'],
        offset=DEFAULT_PARAMS['offset'],
        io_type=DEFAULT_PARAMS['io_type'],
        k=DEFAULT_PARAMS['k'],
        n=DEFAULT_PARAMS['n'],
        **solver_kwargs):
    """`NZ` accepted weight draws and their fixed points for every stimulus of the (bandwidth x contrast x offset) grid
    (ssnode.py:561-590): returns what `find_fixed_points` returns.  Extra keywords go to the solver."""
    solver_options = dict(dict(r0=np.zeros(2 * N)), **solver_kwargs)
    solver_options.update(k=k, n=n, io_type=io_type)
    return find_fixed_points(NZ, _weight_draws(np.random.RandomState(seed), N, J, D, S),
                             _stimulus_rows(N, bandwidths, smoothness, contrast, offset), **solver_options)


def sample_tuning_curves(sample_sites=[0], track_offset_identity=False,
                         include_inhibitory_neurons=False,
                         **kwargs):
    """Tuning curves of the probed neurons of `sample_fixed_points(**kwargs)` (ssnode.py:593-602): returns
    ``(tunings, sample)`` with `tunings` of shape (stimuli [x sites], draws) and `sample` the finder's triple."""
    from .gradient_expressions.utils import subsample_neurons
    sample = sample_fixed_points(**kwargs)
    probed = subsample_neurons(np.array(sample[1]), sample_sites, include_inhibitory_neurons=include_inhibitory_neurons,
                               track_offset_identity=track_offset_identity)
    return probed.T, sample
