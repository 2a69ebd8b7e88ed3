"""``probe_mpsk_snr_est`` -- the SNR probe the reference hangs on frame_equalizer's ``symbols`` port
(``digital.probe_mpsk_snr_est_c(0, 1000, 0.05)``, gnu_radio/IRS_AP.py:275, wired at :312, its ``snr`` message
printed by a message_debug block, gnu_radio/IRS_AP.grc:1607).

Same constructor arguments (``type``, ``msg_nsamples``, ``alpha``), same message ports (``snr``, ``signal``,
``noise``: one double each, every ``msg_nsamples`` samples), same getters.  Two ways in:

* :meth:`update_frame` -- the per-frame moments the HIP chain delivers (``wifirx_out.sym_stats``:
  sum |y|, sum |y|^2, sum |y|^4 over the frame's equalised points), so that the equalised points themselves never have
  to leave the device.  ``wifi_phy_rx(snr_probe=...)`` feeds the probe this way.
* the message port ``in`` (or :meth:`update`) -- 48-point ``carrier`` PDUs, sample by sample, for a flowgraph that
  wires the probe to the block's ``carrier`` port as the reference wires it to ``symbols``.

PROVENANCE.  GNU Radio's gr-digital is not part of the reference tree; the estimators are restated from their published
form as remembered of the 3.10 line (the version the reference names, gnu_radio/IRS_AP.py:10):

* type 0, SNR_EST_SIMPLE: running (cumulative, Welford) mean m and variance v of |y|; signal = m^2, noise = 2 v
  (the radial noise component carries half of the complex noise power), snr = 10 log10(signal / noise).  ``alpha`` is
  not used by this estimator.
* type 2, SNR_EST_M2M4: M2 <- alpha |y|^2 + (1 - alpha) M2, M4 likewise with |y|^4; signal = sqrt(2 M2^2 - M4),
  noise = M2 - signal, snr = 10 log10(signal / noise).
* types 1 (skewness) and 3 (SVR) are not provided.

Per-frame input advances the type-2 averages by whole frames: M <- beta^n M + (1 - beta^n) (frame mean), which is the
sample-by-sample recursion exactly when the statistic is constant over the frame and otherwise weights the frame's
samples equally instead of exponentially; type 0 composes exactly (Chan's merge of means and variances).
"""
from __future__ import annotations

import math

import numpy as np

from . import grshim

SNR_EST_SIMPLE, SNR_EST_SKEW, SNR_EST_M2M4, SNR_EST_SVR = 0, 1, 2, 3


class probe_mpsk_snr_est(grshim.basic_block):
    def __init__(self, type=SNR_EST_SIMPLE, msg_nsamples=10000, alpha=0.001):
        grshim.basic_block.__init__(self, name="probe_mpsk_snr_est", in_sig=None, out_sig=None)
        if type not in (SNR_EST_SIMPLE, SNR_EST_M2M4):
            raise ValueError("probe_mpsk_snr_est: estimator type %r is not provided (0 = simple, 2 = M2M4)" % (type,))
        self._type = int(type)
        self.set_msg_nsamples(msg_nsamples)
        self.set_alpha(alpha)
        self._count = 0                 # samples since the last message
        self._n = 0                     # samples seen (simple estimator)
        self._mean = 0.0
        self._m2agg = 0.0
        self._M2 = 0.0
        self._M4 = 0.0
        for port in ("snr", "signal", "noise"):
            self.message_port_register_out(grshim.intern(port))
        self.message_port_register_in(grshim.intern("in"))
        self.set_msg_handler(grshim.intern("in"), self._handle_pdu)

    # ---- the block's accessors ----
    def type(self):
        return self._type

    def msg_nsample(self):
        return self._nsamples

    msg_nsamples = msg_nsample

    def alpha(self):
        return self._alpha

    def set_msg_nsample(self, n):
        if int(n) <= 0:
            raise ValueError("msg_nsamples must be positive")
        self._nsamples = int(n)

    set_msg_nsamples = set_msg_nsample

    def set_alpha(self, alpha):
        if not (0.0 <= float(alpha) <= 1.0):
            raise ValueError("alpha must lie in [0, 1]")
        self._alpha = float(alpha)
        self._beta = 1.0 - self._alpha

    # ---- estimates ----
    def signal(self):
        if self._type == SNR_EST_SIMPLE:
            return self._mean * self._mean
        return math.sqrt(max(2.0 * self._M2 * self._M2 - self._M4, 0.0))

    def noise(self):
        if self._type == SNR_EST_SIMPLE:
            return 2.0 * self._m2agg / (self._n - 1) if self._n > 1 else 0.0
        return self._M2 - self.signal()

    def snr(self):
        s, n = self.signal(), self.noise()
        if not (s > 0.0 and n > 0.0):
            return float("inf") if s > 0.0 else float("-inf") if n > 0.0 else float("nan")
        return 10.0 * math.log10(s / n)

    # ---- input ----
    def update(self, samples):
        """sample-by-sample recursion over complex points (the `symbols` / `carrier` stream)"""
        y = np.abs(np.asarray(samples, dtype=np.complex128).reshape(-1))
        if self._type == SNR_EST_SIMPLE:
            for a in y:
                self._n += 1
                d = a - self._mean
                self._mean += d / self._n
                self._m2agg += d * (a - self._mean)
        else:
            a, b = self._alpha, self._beta
            for v in y * y:
                self._M2 = a * v + b * self._M2
                self._M4 = a * v * v + b * self._M4
        self._after(len(y))

    def update_frame(self, s1, s2, s4, n):
        """moments of one frame's n equalised points: s1 = sum |y|, s2 = sum |y|^2, s4 = sum |y|^4"""
        n = int(n)
        if n <= 0:
            return
        s1, s2, s4 = float(s1), float(s2), float(s4)
        if self._type == SNR_EST_SIMPLE:
            mb = s1 / n
            m2b = max(s2 - n * mb * mb, 0.0)
            tot = self._n + n
            d = mb - self._mean
            self._mean += d * n / tot
            self._m2agg += m2b + d * d * self._n * n / tot
            self._n = tot
        else:
            w = self._beta ** n
            self._M2 = w * self._M2 + (1.0 - w) * (s2 / n)
            self._M4 = w * self._M4 + (1.0 - w) * (s4 / n)
        self._after(n)

    def _after(self, n):
        self._count += n
        while self._count > self._nsamples:              # upstream's loop: one message per msg_nsamples samples
            self.message_port_pub(grshim.intern("snr"), self.snr())
            self.message_port_pub(grshim.intern("signal"), self.signal())
            self.message_port_pub(grshim.intern("noise"), self.noise())
            self._count -= self._nsamples

    def _handle_pdu(self, msg):
        self.update(grshim.to_python(msg)[-1])
