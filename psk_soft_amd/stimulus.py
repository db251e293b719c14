"""Synthetic PSK stimuli.

``synth_channels`` is the benchmark / parity workload of SURVEY.md section 8(d):
per-channel counter-based PRNG (Philox keyed by 0x5EED0000 + channel), random
M-PSK symbols with a per-channel phase offset, an asymmetric single-peaked
pulse, per-channel gain, a small carrier offset and AWGN.
(The stimulus of the reference's own component test lives with the tests: tests/ref_stimulus.py.)
"""
import math

import numpy as np

SEED_BASE = 0x5EED0000


def pulse_shape(S):
    """Asymmetric single-peaked pulse a_j = 0.2 + 0.8 sin(pi (j+0.9)/(S+1.3)):
    a unique energy maximum per symbol, so the timing argmax has margin."""
    j = np.arange(S, dtype=np.float64)
    return 0.2 + 0.8 * np.sin(np.pi * (j + 0.9) / (S + 1.3))


def synth_channel(channel, M, S, n_complex, sigma=0.01, cfo_max=1e-3, dtype=np.float32, cfo=None, phi0=None):
    """One channel of the section-8(d) workload as interleaved I/Q (length 2*n_complex).
    cfo: M * (carrier phase advance per symbol) given outright instead of drawn from [-cfo_max, cfo_max];
    phi0: the constellation's phase offset given outright (phi0 = 0, cfo = 0: the signal shape of the reference's own test,
    reference tests/test_psk_soft.py:98-117, LinearFit's sums hovering around zero)."""
    g = np.random.Generator(np.random.Philox(key=SEED_BASE + int(channel)))
    n_sym = -(-n_complex // S)
    k = g.integers(0, M, size=n_sym)
    phi0_drawn = g.uniform(0.0, 2 * np.pi / M)
    phi0 = phi0_drawn if phi0 is None else float(phi0)
    gain = g.uniform(0.5, 2.0)
    dphi = g.uniform(-cfo_max, cfo_max) / M  # M * dphi per symbol in [-cfo_max, cfo_max]
    if cfo is not None:
        dphi = float(cfo) / M
    sym_phase = 2 * np.pi * k / M + phi0
    t = np.arange(n_sym * S, dtype=np.float64)
    ph = np.repeat(sym_phase, S) + dphi * (t / S)
    amp = gain * np.tile(pulse_shape(S), n_sym)
    x = amp * np.exp(1j * ph)
    x = x[:n_complex]
    x = x + sigma * (g.standard_normal(n_complex) + 1j * g.standard_normal(n_complex))
    out = np.empty(2 * n_complex, dtype=dtype)
    out[0::2] = x.real
    out[1::2] = x.imag
    return out


def synth_channels(channels, M, S, n_complex, **kw):
    """[len(channels), 2*n_complex] float32; `M` and `S` may be scalars or per-channel lists."""
    channels = list(channels)
    Ms = M if hasattr(M, "__len__") else [M] * len(channels)
    Ss = S if hasattr(S, "__len__") else [S] * len(channels)
    return np.stack([synth_channel(c, Ms[i], Ss[i], n_complex, **kw) for i, c in enumerate(channels)])


def synth_channels_torch(n_channels, M, S, n_complex, device, sigma=0.01, cfo_max=1e-3, seed=SEED_BASE,
                         chunk_channels=256, periodic=False, phase0=False):
    """Same workload, generated on the GPU with torch ops (bench sizes: GiBs of I/Q).
    Returns a [n_channels, 2*n_complex] float32 tensor on `device`.
    periodic=True rounds every channel's carrier offset to the nearest value whose phase advance over
    the buffer is a multiple of 2*pi/M, so that feeding the same buffer again and again (as bench.py
    does) is one continuous stream for the carrier loop instead of a phase jump per call.
    phase0=True: every channel's constellation sits at zero phase with no carrier offset -- the signal
    shape of the reference's own component test (reference tests/test_psk_soft.py:98-117: ideal
    constellation points, no offset), with this workload's pulse, gains and noise.  The M-th-power phase
    is then noise around zero and LinearFit's running sums hover around zero in every channel."""
    import torch

    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))
    out = torch.empty((n_channels, 2 * n_complex), dtype=torch.float32, device=device)
    n_sym = -(-n_complex // S)
    pulse = torch.tensor(pulse_shape(S), dtype=torch.float32, device=device)
    for c0 in range(0, n_channels, chunk_channels):
        c1 = min(n_channels, c0 + chunk_channels)
        nc = c1 - c0
        k = torch.randint(0, M, (nc, n_sym), generator=gen, device=device)
        phi0 = torch.rand((nc, 1), generator=gen, device=device) * (2 * math.pi / M)
        gain = 0.5 + 1.5 * torch.rand((nc, 1), generator=gen, device=device)
        dphi = (2 * torch.rand((nc, 1), generator=gen, device=device) - 1) * (cfo_max / M)
        if phase0:
            phi0 = phi0 * 0.0
            dphi = dphi * 0.0
        if periodic and n_complex % S == 0:
            q = 2 * math.pi / (M * n_sym)
            dphi = torch.round(dphi / q) * q
        tsym = torch.arange(n_sym, device=device, dtype=torch.float32).unsqueeze(0)
        sym_phase = k.to(torch.float32) * (2 * math.pi / M) + phi0 + dphi * tsym
        # CFO ramp inside a symbol is < 1e-3/M rad: applied per symbol (documented approximation)
        re = (torch.cos(sym_phase) * gain).unsqueeze(2) * pulse
        im = (torch.sin(sym_phase) * gain).unsqueeze(2) * pulse
        re = re.reshape(nc, n_sym * S)[:, :n_complex]
        im = im.reshape(nc, n_sym * S)[:, :n_complex]
        view = out[c0:c1].view(nc, n_complex, 2)
        view[:, :, 0] = re + sigma * torch.randn((nc, n_complex), generator=gen, device=device)
        view[:, :, 1] = im + sigma * torch.randn((nc, n_complex), generator=gen, device=device)
        del k, re, im, sym_phase
    return out
