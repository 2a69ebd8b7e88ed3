"""``wifi_phy_rx`` -- drop-in for the RX half of the reference's ``wifi_phy_hier`` hier block.

Port names, parameters and setters are those of gnu_radio/wifi_phy_hier.grc (pad ``samp_in``
:587-604, message pads ``mac_out`` / ``carrier`` :605-640, parameters ``bandwidth``, ``chan_est``,
``encoding``, ``frequency``, ``sensitivity`` :83-92,299-308,442-451,501-510,681-690); inside
IRS_AP it replaces the inlined blocks gnu_radio/IRS_AP.py:268-269,271-273,276-285 and feeds the
consumers wired at :291-293.  ``work()`` hands the complex64 chunk through the ctypes C ABI to
the HIP chain (wifirx_push) and publishes what wifirx_poll returns:

* ``mac_out``: ``(meta, u8vector)`` with the MAC frame *without* FCS, starting at the 24-byte MAC
  header (what ``decode_mac`` publishes; the consumer strips ``[24:][4:]``,
  gnu_radio/IRS_AP_epy_block_2.py:31-36); ``meta`` carries ``frame_bytes, encoding, snr, freq,
  freq_offset, dlt``.
* ``carrier``: ``({}, c32vector[48])`` per data symbol (``frame_equalizer.symbols``,
  gnu_radio/IRS_AP.py:293).

Frames with a bad SIGNAL parity or FCS are dropped silently, as in the reference; ``work()``
never raises for channel conditions.
"""
from __future__ import annotations

import math

import numpy as np

from . import capi, grshim, probe

LS, LMS, COMB, STA = 0, 1, 2, 3
LINKTYPE_IEEE802_11 = 105


class wifi_phy_rx(grshim.sync_block):
    def __init__(self, bandwidth=10e6, chan_est=LS, encoding=0, frequency=5.89e9, sensitivity=0.56,
                 max_sym=511, publish_carrier=True, device=0, batch_samples=1 << 20, publish_csi=False, snr_probe=None):
        grshim.sync_block.__init__(self, name="wifi_phy_rx", in_sig=[np.complex64], out_sig=None)
        self.bandwidth = float(bandwidth)
        self.chan_est = int(chan_est)
        self.encoding = encoding            # TX-side parameter of the hier block: accepted, unused on RX
        self.frequency = float(frequency)
        self.sensitivity = float(sensitivity)
        self.publish_carrier = bool(publish_carrier)
        self.publish_csi = bool(publish_csi)      # "csi" entry of the mac_out dictionary, as upstream's frame_equalizer tags it
        # snr_probe = (type, msg_nsamples, alpha) or True for the reference's (0, 1000, 0.05): the
        # digital.probe_mpsk_snr_est_c of gnu_radio/IRS_AP.py:275,312, fed with per-frame moments from the device
        # (no equalised points cross the bus); its `snr` / `signal` / `noise` ports are self.snr_probe's
        self.snr_probe = None
        if snr_probe:
            args = (0, 1000, 0.05) if snr_probe is True else tuple(snr_probe)
            self.snr_probe = probe.probe_mpsk_snr_est(*args)
        self.message_port_register_out(grshim.intern("mac_out"))
        self.message_port_register_out(grshim.intern("carrier"))
        self._rx = capi.WifiRx(bandwidth=self.bandwidth, frequency=self.frequency, sensitivity=self.sensitivity,
                               chan_est=self.chan_est, max_sym=max_sym, llr_bits=0,
                               want_carrier=self.publish_carrier, device=device)
        # The scheduler calls work() with a few thousand items; the GPU pipeline runs once `batch_samples` have come in
        # (about 50 ms of signal at 20 MS/s) -- one device round trip per work() call could not keep up -- and at stop().
        self.batch_samples = int(batch_samples)
        self._rx.set_param(capi.P_STREAM_BATCH, self.batch_samples)
        self._pend, self._pend_n = [], 0
        self.frames_ok = 0
        self.frames_dropped = 0

    # ---- setters generated for the hier block's parameters (gnu_radio/IRS_user.py:229,265,273;
    #      equaliser setters at gnu_radio/IRS_AP.py:348,373,382) ----
    def get_bandwidth(self):
        return self.bandwidth

    def set_bandwidth(self, bandwidth):
        self.bandwidth = float(bandwidth)
        self._rx.set_param(capi.P_BANDWIDTH, self.bandwidth)

    def get_frequency(self):
        return self.frequency

    def set_frequency(self, frequency):
        self.frequency = float(frequency)
        self._rx.set_param(capi.P_FREQUENCY, self.frequency)

    def get_chan_est(self):
        return self.chan_est

    def set_chan_est(self, chan_est):
        self._rx.set_param(capi.P_CHAN_EST, int(chan_est))     # LS, LMS, COMB, STA
        self.chan_est = int(chan_est)

    set_algorithm = set_chan_est                                # frame_equalizer.set_algorithm

    def get_sensitivity(self):
        return self.sensitivity

    def set_sensitivity(self, sensitivity):
        self.sensitivity = float(sensitivity)
        self._rx.set_param(capi.P_SENSITIVITY, self.sensitivity)

    def get_encoding(self):
        return self.encoding

    def set_encoding(self, encoding):
        self.encoding = encoding

    # ---- stream interface ----
    def work(self, input_items, output_items):
        x = input_items[0]
        n = len(x)
        if n >= self.batch_samples and not self._pend:
            self._rx.push(x)                              # a chunk that is a batch by itself goes straight through
            self._publish()
        elif n:
            # the scheduler owns the input buffer: keep a copy and hand the library one piece per batch
            self._pend.append(np.array(x, dtype=np.complex64, copy=True))
            self._pend_n += n
            if self._pend_n >= self.batch_samples:
                self._flush_pending()
                self._publish()
        return n

    def _flush_pending(self):
        if self._pend:
            self._rx.push(self._pend[0] if len(self._pend) == 1 else np.concatenate(self._pend))
            self._pend, self._pend_n = [], 0

    def stop(self):
        """End of stream: settle the frames still waiting for samples."""
        self._flush_pending()
        self._rx.push(np.zeros(0, dtype=np.complex64))
        self._publish()
        return True

    def _publish(self):
        while True:
            r = self._rx.poll(cap=64, psdu_stride=2048, want_csi=self.publish_csi, want_stats=self.snr_probe is not None)
            fr = r["frames"]
            if len(fr) == 0:
                return
            for i in range(len(fr)):
                f = fr[i]
                if self.snr_probe is not None and int(f["n_sym_out"]) > 0:
                    st = r["sym_stats"][i]
                    self.snr_probe.update_frame(st[0], st[1], st[2], 48 * int(f["n_sym_out"]))
                if self.publish_carrier and r["carrier"] is not None:
                    for s in range(int(f["n_sym_out"])):
                        self.message_port_pub(grshim.intern("carrier"), grshim.make_pdu({}, r["carrier"][i, s]))
                if not (int(f["flags"]) & capi.F_CRC_OK):
                    self.frames_dropped += 1
                    continue
                self.frames_ok += 1
                tag = float(np.float64(f["cfo_coarse"]) - np.float64(f["cfo_fine"]))
                meta = {
                    "frame_bytes": int(f["psdu_len"]),
                    "encoding": int(f["encoding"]),
                    "snr": float(f["snr_db"]),
                    "freq": self.frequency,
                    "freq_offset": tag * self.bandwidth / (2 * math.pi),
                    "dlt": LINKTYPE_IEEE802_11,
                }
                if self.publish_csi:
                    meta["csi"] = r["csi"][i].copy()
                blob = r["psdu"][i, :int(f["psdu_len"]) - 4].copy()
                self.message_port_pub(grshim.intern("mac_out"), grshim.make_pdu(meta, blob))

    def get_probe_snr(self):
        """latest estimate of the SNR probe in dB (None without a probe)"""
        return None if self.snr_probe is None else self.snr_probe.snr()

    def stats(self):
        return self._rx.stats()

    def close(self):
        """Release the library handle (device buffers, stream)."""
        self._rx.close()
