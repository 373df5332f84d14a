"""The device-resident tracking optimiser (gslam_amd/csrc/track_opt.h) against torch.optim.Adam + torch.optim.LBFGS,
the host logic it replaces (gslam/frontend.py:604-662).  The state machine is plain C: here it is compiled for the
host with gcc and driven with analytic objectives; the GPU test runs the same code through the C ABI."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def sm():
    out = os.path.join(HERE, "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libtrackopt_host.so")
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-ffp-contract=off", "-o", so,
                           os.path.join(HERE, "trackopt_host.c"), "-lm"])
    lib = C.CDLL(so)
    lib.trackopt_state_bytes.restype = C.c_long
    lib.trackopt_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_double, C.c_int, C.c_int, C.c_int,
                                  C.c_double, C.c_double]
    lib.trackopt_advance.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]
    lib.trackopt_loss.restype = C.c_double
    return lib


def rosenbrock(x):
    return (100.0 * (x[1:] - x[:-1] ** 2) ** 2 + (1.0 - x[:-1]) ** 2).sum()


def quadratic(x):
    n = x.numel()
    A = torch.diag(torch.linspace(1.0, 30.0, n, dtype=x.dtype)) + 0.3
    b = torch.linspace(-1.0, 1.0, n, dtype=x.dtype)
    return 0.5 * x @ (A @ x) - b @ x + 0.1 * torch.cos(3.0 * x).sum()


def tracking_like(x):
    # smooth, badly scaled, with a flat direction: the shape of the photometric pose objective
    s = torch.tensor([1.0, 1.0, 1.0, 40.0, 40.0, 40.0, 40.0, 40.0, 40.0, 0.1, 0.1], dtype=x.dtype)[: x.numel()]
    r = torch.sin(s * x) + 0.05 * s * x
    return (r ** 2).mean() + 1e-3 * torch.tanh(x).sum()


def run_torch(fn, x0, n_adam, lr, max_eval=25, history=5):
    x = x0.clone().requires_grad_(True)
    trace = []

    def closure():
        if x.grad is not None:
            x.grad = None
        loss = fn(x)
        loss.backward()
        trace.append((x.detach().clone(), float(loss.detach())))
        return loss

    adam = torch.optim.Adam([x], lr)
    for _ in range(n_adam):
        closure()
        adam.step()
    opt = torch.optim.LBFGS([x], history_size=history, line_search_fn="strong_wolfe", tolerance_change=1e-9, lr=lr,
                            max_eval=max_eval)
    opt.step(closure)
    return x.detach(), trace


def run_sm(lib, fn, x0, n_adam, lr, max_eval=25, history=5, extra=0):
    n = x0.numel()
    st = C.create_string_buffer(lib.trackopt_state_bytes())
    lib.trackopt_init(st, n, n_adam, lr, lr, history, 20, max_eval, 1e-7, 1e-9)
    p = x0.clone().numpy().astype(np.float32)
    trace = []
    n_calls = 0
    while lib.trackopt_phase(st) != 4 or n_calls < extra:
        x = torch.from_numpy(p.copy()).requires_grad_(True)
        loss = fn(x)
        loss.backward()
        g = x.grad.numpy().astype(np.float32)
        if lib.trackopt_phase(st) != 4:
            trace.append((torch.from_numpy(p.copy()), float(loss)))
        lib.trackopt_advance(st, p.ctypes.data, g.ctypes.data, float(loss))
        n_calls += 1
        assert n_calls < 200
    return torch.from_numpy(p.copy()), trace, st


@pytest.mark.parametrize("fn,n,n_adam,lr", [
    (rosenbrock, 2, 0, 1.0), (rosenbrock, 6, 0, 1.0), (rosenbrock, 11, 0, 1.0), (quadratic, 11, 0, 1.0),
    (quadratic, 5, 0, 0.5),
])
def test_state_machine_follows_torch(sm, fn, n, n_adam, lr):
    """Well-conditioned line searches: the machine proposes the same evaluation points as torch.optim.LBFGS, one
    for one (bracketing, zoom, history update, two-loop recursion, stopping rules), until both sit at the minimum
    and float32 noise in the loss decides."""
    g = torch.Generator().manual_seed(n * 7 + n_adam)
    x0 = (torch.rand(n, generator=g) - 0.5).float()
    xt, tt = run_torch(fn, x0, n_adam, lr)
    xs, ts, st = run_sm(sm, fn, x0, n_adam, lr)
    fmin = min(f for _, f in tt)
    compared = 0
    for k, ((xa, fa), (xb, fb)) in enumerate(zip(ts, tt)):
        if k > 0 and abs(tt[k - 1][1] - fmin) <= 2e-6 * max(1.0, abs(fmin)):
            break                                   # converged: the remaining decisions are rounding noise
        assert torch.allclose(xa, xb, rtol=1e-3, atol=1e-5), (k, xa, xb)
        assert abs(fa - fb) <= 3e-3 * max(1.0, abs(fb)), (k, fa, fb)   # torch interpolates in f32, the machine in f64
        compared += 1
    assert compared >= min(len(tt), 8), compared
    assert abs(len(ts) - len(tt)) <= 3, (len(ts), len(tt))
    assert abs(float(fn(xs)) - float(fn(xt))) <= 1e-4 * max(1.0, abs(float(fn(xt))))


@pytest.mark.parametrize("fn,n,n_adam,lr", [
    (quadratic, 11, 10, 0.002), (tracking_like, 11, 10, 0.002), (tracking_like, 11, 0, 0.002),
    (rosenbrock, 11, 10, 0.01),
])
def test_tracker_regime(sm, fn, n, n_adam, lr):
    """The tracker's regime (lr = 2e-3: tiny first steps, objective almost linear between trial points).  There the
    branch of the cubic interpolation hangs on the last bit of the float32 loss, in torch as well, so trajectories
    are compared up to the first such coin flip and the outcome statistically: identical Adam phase, the same budget
    of evaluations, monotone progress, a final loss of the same order."""
    g = torch.Generator().manual_seed(n * 7 + n_adam)
    x0 = (torch.rand(n, generator=g) - 0.5).float()
    xt, tt = run_torch(fn, x0, n_adam, lr)
    xs, ts, st = run_sm(sm, fn, x0, n_adam, lr)
    assert len(ts) == len(tt), (len(ts), len(tt))
    for k in range(n_adam + 3):                      # Adam steps + the first L-BFGS evaluations
        assert torch.allclose(ts[k][0], tt[k][0], rtol=1e-4, atol=1e-6), k
    f0, fs, ft = ts[0][1], float(fn(xs)), float(fn(xt))
    assert fs < f0 and ft < f0
    assert fs <= 3.0 * ft + 1e-6 and ft <= 3.0 * fs + 1e-6, (fs, ft)


def test_done_is_sticky(sm):
    x0 = torch.tensor([0.3, -0.2, 0.1])
    xs, ts, st = run_sm(sm, rosenbrock, x0, 0, 1.0)
    n_evals = sm.trackopt_evals(st)
    xs2, _, st2 = run_sm(sm, rosenbrock, x0, 0, 1.0, extra=len(ts) + 5)
    assert torch.equal(xs, xs2) and sm.trackopt_evals(st2) == n_evals
    assert sm.trackopt_stop_reason(st) in (1, 2, 3, 4, 5, 6, 7)


def test_respects_max_eval(sm):
    x0 = (torch.arange(8).float() / 8.0 - 0.5)
    for max_eval in (5, 12, 25):
        _, tt = run_torch(rosenbrock, x0, 0, 1.0, max_eval=max_eval)
        _, ts, _ = run_sm(sm, rosenbrock, x0, 0, 1.0, max_eval=max_eval)
        assert len(ts) == len(tt)
