"""A structurally independent float64 NumPy receiver (test infrastructure).

Written to SURVEY.md Appendix A, literally, for the LS equaliser: complex128 throughout, `np.fft.fft`, `np.angle`,
`np.exp`, complex division.  It shares NOTHING with the product or the oracle: no `include/wifirx_tables.h`, no
`oracle/`, no `wifirx.*` import.  Its constants come from the reference's own flowgraph as extracted into
`tests/golden/grc_constants.json` (carrier map, pilot polarity, the long training symbol = sync word 3) and from IEEE
802.11 (rate field, convolutional code, constellations).  Its purpose is to catch a slip that the oracle's two modes
(SPEC / LIBM: same file, same hand, same tables) would share -- a wrong sign, index, cadence or constant.

Batch semantics as everywhere in this repository: one frame per slot, the slot's past and future are zeros, the first
trigger of sync_short counts.

    rx = IndependentRx()                       # constants from tests/golden/grc_constants.json
    out = rx.receive(slots)                    # slots: [F, S] complex; dict of per-frame arrays
"""
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# IEEE 802.11-2012 Table 18-6: RATE field R1..R4 (R1 first = bit 0) -> (N_BPSC, N_DBPS); value = bits LSB-first
RATE_FIELD = {0b1011: (0, 1, 24), 0b1111: (1, 1, 36), 0b1010: (2, 2, 48), 0b1110: (3, 2, 72),
              0b1001: (4, 4, 96), 0b1101: (5, 4, 144), 0b1000: (6, 6, 192), 0b1100: (7, 6, 216)}
MIN_GAP, MAX_SAMPLES, SYNC_LENGTH = 480, 540 * 80, 320


def _conv_outputs():
    """K = 7 code of 18.3.5.6: A = x0+x2+x3+x5+x6, B = x0+x1+x2+x3+x6 (x_d = input d steps ago).  State = the six previous
    inputs, newest in bit 0.  Returns for every (state, input): next state, output pair."""
    nxt = np.zeros((64, 2), np.int64)
    out = np.zeros((64, 2, 2), np.int64)
    for st in range(64):
        past = [(st >> d) & 1 for d in range(6)]          # past[d-1] = x_d
        for b in (0, 1):
            x = [b] + past
            out[st, b, 0] = x[0] ^ x[2] ^ x[3] ^ x[5] ^ x[6]
            out[st, b, 1] = x[0] ^ x[1] ^ x[2] ^ x[3] ^ x[6]
            nxt[st, b] = ((st << 1) | b) & 63
    return nxt, out


class IndependentRx:
    def __init__(self, bandwidth=20e6, frequency=5.89e9, threshold=0.56, min_plateau=2):
        with open(os.path.join(GOLD, "grc_constants.json")) as f:
            g = json.load(f)
        assert g["fft_len"] == 64 and g["cp_len"] == 16 and g["sync_length"] == SYNC_LENGTH
        self.bw, self.fc, self.thr, self.min_plateau = float(bandwidth), float(frequency), float(threshold), int(min_plateau)
        self.win_a, self.win_p = int(g["window_size"]), int(g["window_size"]) + 16
        self.data_bins = np.array(g["occupied_carriers"][0]) + 32             # shifted index i = k + 32
        self.pilot_bins = np.array(g["pilot_carriers"][0]) + 32
        assert list(self.pilot_bins) == [11, 25, 39, 53] and len(self.data_bins) == 48
        ps = np.array(g["pilot_symbols"], dtype=np.float64)
        assert ps.shape == (127, 4)
        self.polarity = ps[:, 0]                                              # (p, p, p, -p) per symbol
        assert np.array_equal(ps[:, 1], ps[:, 0]) and np.array_equal(ps[:, 3], -ps[:, 0])
        lts_f = np.array([complex(a, b) for a, b in g["sync_words"][3]])      # L_k at shifted index k + 32
        assert np.allclose(lts_f.imag, 0) and lts_f.shape == (64,)
        self.lts_f = lts_f.real
        self.used = np.abs(self.lts_f) > 0.5
        assert self.used.sum() == 52 and not self.used[32]
        self.lts_t = np.fft.ifft(np.fft.ifftshift(lts_f)) * 64.0               # the long training symbol in time (any scale)
        self.nxt, self.cout = _conv_outputs()

    # ---- A.1 + A.2 ------------------------------------------------------------------------------------------
    def detect(self, x):
        F, S = x.shape
        xd = np.concatenate([np.zeros((F, 16), x.dtype), x[:, :-16]], axis=1)           # delay(16)
        a = x * np.conj(xd)
        ca = np.concatenate([np.zeros((F, 1), a.dtype), np.cumsum(a, axis=1)], axis=1)
        p = np.abs(x) ** 2
        cp = np.concatenate([np.zeros((F, 1)), np.cumsum(p, axis=1)], axis=1)
        n = np.arange(S)
        A = ca[:, n + 1] - ca[:, np.maximum(n + 1 - self.win_a, 0)]                     # moving_average_cc(48)
        P = cp[:, n + 1] - cp[:, np.maximum(n + 1 - self.win_p, 0)]                     # moving_average_ff(64)
        with np.errstate(divide="ignore", invalid="ignore"):
            c = np.abs(A) / P
        above = c > self.thr                                                            # NaN compares false
        run = above.copy()
        for j in range(1, self.min_plateau + 1):                                        # min_plateau + 1 consecutive samples
            run[:, j:] &= above[:, :-j]
            run[:, :j] = False
        has = run.any(axis=1)
        t = np.where(has, run.argmax(axis=1), -1)
        cfo = np.where(has, np.angle(A[np.arange(F), np.maximum(t, 0)]) / 16.0, 0.0)
        return t, cfo

    # ---- A.3 ------------------------------------------------------------------------------------------------
    def _copied(self, x, t, cfo_c, m0, n):
        """y[m] = x[t - 16 + m] exp(-j cfo_c m) for m = m0 .. m0 + n - 1 (zeros outside the slot); m0 per frame"""
        F, S = x.shape
        m = m0[:, None] + np.arange(n)[None, :]
        idx = t[:, None] - 16 + m
        ok = (idx >= 0) & (idx < S)
        v = np.where(ok, x[np.arange(F)[:, None], np.clip(idx, 0, S - 1)], 0.0)
        return v * np.exp(-1j * cfo_c[:, None] * m), m

    def sync_long(self, x, t, cfo_c, L):
        F = x.shape[0]
        y, _ = self._copied(x, t, cfo_c, np.zeros(F, np.int64), SYNC_LENGTH + 64)
        win = np.lib.stride_tricks.sliding_window_view(y, 64, axis=1)[:, :SYNC_LENGTH, :]     # [F, 320, 64]
        corr = win @ np.conj(self.lts_t)
        mag = np.abs(corr)
        top = np.argsort(-mag, axis=1, kind="stable")[:, :4]                    # ties -> lower lag first
        fs = np.full(F, SYNC_LENGTH)
        cfo_f = np.zeros(F)
        found = np.zeros(F, np.int64)
        active = L >= SYNC_LENGTH + 63
        rows = np.arange(F)
        for i in range(3):
            for k in range(i + 1, 4):
                oi, ok_ = top[:, i], top[:, k]
                lo, hi = np.minimum(oi, ok_), np.maximum(oi, ok_)
                diff = hi - lo
                hit = active & ((diff == 63) | (diff == 64) | (diff == 65))
                ang = np.angle(corr[rows, lo] * np.conj(corr[rows, hi]))        # first = the earlier peak
                fs = np.where(hit, lo, fs)
                cfo_f = np.where(hit, ang / np.maximum(diff, 1), cfo_f)
                found = np.where(hit, diff, found)
                active = active & ~(hit & (diff == 64))                         # a distance of 64 ends the search
        return fs, cfo_f, found, top

    # ---- A.5 (7) --------------------------------------------------------------------------------------------
    def viterbi24(self, coded):
        """coded [F, 48] hard bits (A0 B0 A1 B1 ...) -> [F, 24] decoded bits; start state 0, best end state"""
        F = coded.shape[0]
        pm = np.full((F, 64), 1 << 20, np.int64)
        pm[:, 0] = 0
        back = np.zeros((24, F, 64), np.int64)
        prev_of = [[(st, b) for st in range(64) for b in (0, 1) if self.nxt[st, b] == ns] for ns in range(64)]
        for tt in range(24):
            ra, rb = coded[:, 2 * tt], coded[:, 2 * tt + 1]
            new = np.empty_like(pm)
            for ns in range(64):
                (s0, b0), (s1, b1) = prev_of[ns]
                m0 = pm[:, s0] + (ra != self.cout[s0, b0, 0]) + (rb != self.cout[s0, b0, 1])
                m1 = pm[:, s1] + (ra != self.cout[s1, b1, 0]) + (rb != self.cout[s1, b1, 1])
                take1 = m1 < m0
                new[:, ns] = np.where(take1, m1, m0)
                back[tt, :, ns] = np.where(take1, s1, s0)
            pm = new
        st = pm.argmin(axis=1)
        bits = np.zeros((F, 24), np.int64)
        rows = np.arange(F)
        for tt in range(23, -1, -1):
            bits[:, tt] = st & 1                                                 # newest input sits in bit 0 of the state
            st = back[tt, rows, st]
        return bits

    @staticmethod
    def decide(y, n_bpsc):
        """A.6: LSB = first transmitted bit"""
        re, im = y.real, y.imag
        if n_bpsc == 1:
            return (re > 0).astype(np.uint8)
        if n_bpsc == 2:
            return ((re > 0) | ((im > 0) << 1)).astype(np.uint8)
        if n_bpsc == 4:
            a = np.sqrt(0.1)
            return ((re > 0) | ((np.abs(re) < 2 * a) << 1) | ((im > 0) << 2) | ((np.abs(im) < 2 * a) << 3)).astype(np.uint8)
        a = np.sqrt(1.0 / 42.0)
        ar, ai = np.abs(re), np.abs(im)
        return ((re > 0) | ((ar < 4 * a) << 1) | (((ar > 2 * a) & (ar < 6 * a)) << 2) |
                ((im > 0) << 3) | ((ai < 4 * a) << 4) | (((ai > 2 * a) & (ai < 6 * a)) << 5)).astype(np.uint8)

    # ---- the chain ------------------------------------------------------------------------------------------
    def receive(self, slots, max_sym=64):
        x = np.asarray(slots, dtype=np.complex128)
        F, S = x.shape
        t, cfo_c = self.detect(x)
        det = t >= 0
        tt = np.where(det, t, 16)
        L = np.where(det, np.minimum(S - (tt - 16), MAX_SAMPLES), 0)
        fs, cfo_f, found, top = self.sync_long(x, tt, cfo_c, L)
        sync = det & (found > 0)
        tag = cfo_c - cfo_f                                                      # sync_long's wifi_start tag
        eps0 = tag * self.bw / (2 * np.pi * self.fc)
        d_er = np.zeros(F)
        prev = np.zeros((F, 4), np.complex128)
        H = np.ones((F, 64), np.complex128)
        alive = sync.copy()
        n_sym = np.zeros(F, np.int64)
        n_bpsc = np.ones(F, np.int64)
        enc = np.zeros(F, np.int64)
        length = np.zeros(F, np.int64)
        signal_ok = np.zeros(F, bool)
        truncated = np.zeros(F, bool)
        snr = np.zeros(F)
        idx = np.zeros((F, max_sym, 48), np.uint8)
        n_out = np.zeros(F, np.int64)
        k64 = np.arange(64) - 32
        s = 0
        while True:
            act = alive & (s <= n_sym + 2)
            if not act.any():
                break
            off0 = fs + (64 * s if s < 2 else 128 + 80 * (s - 2) + 16)
            short = act & ((off0 + 64 > L) | ((s > 2) & (s - 3 >= max_sym)))
            truncated |= short
            alive &= ~short
            act &= ~short
            if not act.any():
                break
            y, m = self._copied(x, tt, cfo_c, off0, 64)                          # sync_short's copy ...
            z = y * np.exp(1j * m * cfo_f[:, None])                              # ... and sync_long's
            X = np.fft.fftshift(np.fft.fft(z, axis=1), axes=1)                   # A.4: bin 0 <-> k = -32
            X = X * np.exp(1j * 2 * np.pi * s * 80 * (eps0 + d_er)[:, None] * k64[None, :] / 64.0)          # (1)
            X11, X25, X39, X53 = X[:, 11], X[:, 25], X[:, 39], X[:, 53]
            if s < 2:                                                                                       # (2), (3)
                beta = np.angle(X11 - X25 + X39 + X53)
                cur = np.stack([X11, -X25, X39, X53], axis=1)
            else:
                p = self.polarity[(s - 2) % 127]
                beta = np.angle(p * (X11 + X25 + X39 - X53))
                cur = p * np.stack([X11, X25, X39, -X53], axis=1)
                er_new = np.angle((np.conj(prev) * cur).sum(axis=1)) * self.bw / (2 * np.pi * self.fc * 80)
            prev = np.where(act[:, None], cur, prev)
            X = X * np.exp(-1j * beta)[:, None]                                                             # (4)
            if s >= 2:
                d_er = np.where(act, 0.9 * d_er + 0.1 * er_new, d_er)                                       # (5)
            if s == 0:                                                                                      # (6)
                H = np.where(act[:, None], X, H)
            elif s == 1:
                u = self.used
                noise = (np.abs(H[:, u] - X[:, u]) ** 2).sum(axis=1)
                signal = (np.abs(H[:, u] + X[:, u]) ** 2).sum(axis=1)
                Hn = np.ones_like(H)
                Hn[:, u] = (H[:, u] + X[:, u]) / (2.0 * self.lts_f[u])[None, :]
                H = np.where(act[:, None], Hn, H)
                with np.errstate(divide="ignore", invalid="ignore"):
                    snr = np.where(act, 10 * np.log10(signal / noise / 2), snr)
            else:
                Y = X[:, self.data_bins] / H[:, self.data_bins]
                if s == 2:                                                                                  # (7)
                    rxb = (Y.real > 0).astype(np.int64)
                    i48 = np.arange(48)
                    deint = rxb[:, 3 * (i48 % 16) + i48 // 16]
                    bits = self.viterbi24(deint)
                    w = 1 << np.arange(24)
                    val = (bits * w).sum(axis=1)
                    parity = bits[:, :17].sum(axis=1) & 1
                    rate = val & 15
                    ln = (val >> 5) & 0xfff
                    ok = act & (parity == bits[:, 17]) & np.isin(rate, list(RATE_FIELD))
                    for r, (e, nb, nd) in RATE_FIELD.items():
                        sel = ok & (rate == r)
                        enc[sel], n_bpsc[sel] = e, nb
                        n_sym[sel] = -(-(16 + 8 * ln[sel] + 6) // nd)
                    length = np.where(ok, ln, length)
                    signal_ok |= ok
                    alive &= ~(act & ~ok)
                else:                                                                                       # (8)
                    q = s - 3
                    for nb in (1, 2, 4, 6):
                        sel = act & (n_bpsc == nb)
                        if sel.any():
                            idx[sel, q, :] = self.decide(Y[sel], nb)
                    n_out = np.where(act, q + 1, n_out)
            s += 1
        complete = signal_ok & (n_out == n_sym) & ~truncated
        return dict(trigger=t, cfo_coarse=cfo_c, frame_start=np.where(sync, fs, 0), cfo_fine=np.where(sync, cfo_f, 0.0),
                    detected=det, sync=sync, found=found, top4=top, signal_ok=signal_ok, encoding=enc, psdu_len=length,
                    n_sym=n_sym, n_sym_out=n_out, complete=complete, snr_db=snr, idx=idx)
