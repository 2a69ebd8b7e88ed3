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

import ctypes
import math

import numpy as np

from . import capi, grshim, probe

LS, LMS, COMB, STA = 0, 1, 2, 3
_C64 = np.dtype(np.complex64)
_ANCHOR = ctypes.c_char * 1
LINKTYPE_IEEE802_11 = 105


class wifi_phy_rx(grshim.sync_block):
    def __init__(self, bandwidth=10e6, chan_est=LS, encoding=0, frequency=5.89e9, sensitivity=0.56,
                 max_sym=511, publish_carrier=True, device=0, batch_samples=None, publish_csi=False, snr_probe=None):
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
        # The scheduler calls work() with a few thousand items.  The library copies them into a pinned staging buffer
        # and returns; once `batch_samples` have come in its worker thread runs the device pipeline for that batch while
        # work() keeps filling the next one -- one device round trip per work() call could not keep up.  Default: 50 ms
        # of signal at the block's bandwidth (2^16 samples at least), so that a PDU leaves `mac_out` about 50-100 ms
        # after its last sample arrived (the reference's decode_mac publishes per frame; INTEGRATION.md, "Latency"); a
        # file-driven run that only cares for throughput passes a larger value (bench.py: 2^22).  Finished frames are
        # published by the first work() call after their batch is through (wifirx_queued: one atomic load per call), and
        # at stop().
        if batch_samples is None:
            batch_samples = max(1 << 16, int(0.05 * self.bandwidth))
        self.batch_samples = min(int(batch_samples), capi.STREAM_BATCH_MAX)
        self._rx.set_param(capi.P_STREAM_BATCH, self.batch_samples)
        self._rx.set_param(capi.P_STREAM_IDX, 0)            # PDUs only: the hard decisions stay on the device
        self._push = capi.lib().wifirx_push
        self._queued = capi.lib().wifirx_queued
        self._h = self._rx._h
        self._p_mac = grshim.intern("mac_out")
        self._p_car = grshim.intern("carrier")
        self.frames_ok = 0
        self.frames_dropped = 0
        self.raise_on_error = True
        self.push_errors = 0
        self.last_error = ""
        self.stream_dead = False        # the library has declared the stream dead (WIFIRX_EDEAD): work() ends the block

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
        if n:
            if x.dtype != _C64 or not x.flags.c_contiguous:
                x = np.ascontiguousarray(x, dtype=np.complex64)
            try:
                buf = _ANCHOR.from_buffer(x)                 # the address of x's data, without building a dict
            except (TypeError, ValueError):                  # read-only input: the slower route
                buf = x.ctypes.data
            rc = self._push(self._h, buf, n, 0)              # copies; x is the scheduler's again
            if rc:
                # Nothing is lost or doubled: what the library took over (usually nothing) counts as consumed, the
                # scheduler hands the rest in again with the next call (include/wifirx.h, WIFIRX_P_STREAM_BATCH: ERRORS).
                # raise_on_error (default): the error surfaces as an exception of the block, as any failing GNU Radio
                # block's would; otherwise it is counted and work() reports the consumed items.
                self.push_errors += 1
                self.last_error = capi.lib().wifirx_last_error(self._h).decode()
                if rc == capi.EDEAD:
                    # the stream is gone for good (include/wifirx.h: ERRORS): returning "0 items consumed" would have the
                    # scheduler hand the same items in for ever.  Publish what is finished, then end the block -- by the
                    # exception with raise_on_error, by WORK_DONE (-1) without.
                    self.stream_dead = True
                    self._publish()
                    if self.raise_on_error:
                        self._rx._check(rc)
                    return -1
                if self.raise_on_error:
                    self._rx._check(rc)
                n = self._rx.push_consumed()
            if self._queued(self._h):
                self._publish()
        return n

    def stop(self):
        """End of stream: settle the frames still waiting for samples.  A batch that failed on the library's worker thread is
        reported by the first flush (once, before it does anything) and run again by the second: flush until it goes through
        (at most three times), publish whatever is finished in any case, and only then -- with raise_on_error -- raise."""
        rc = 0
        try:
            for _ in range(3):
                if self.stream_dead:                 # no flush can succeed any more (work() has reported it)
                    break
                rc = self._push(self._h, None, 0, 0)
                if rc == 0:
                    break
                self.push_errors += 1
                self.last_error = capi.lib().wifirx_last_error(self._h).decode()
                if rc == capi.EDEAD:
                    self.stream_dead = True
        finally:
            self._publish()
        if rc and self.raise_on_error:
            self._rx._check(rc)
        return True

    def _publish(self):
        """Everything the library has finished so far, as PDUs.  Per-frame fields are taken out of the record array
        column by column (one tolist() per field, not a NumPy scalar conversion per frame and field)."""
        pub, make = self.message_port_pub, grshim.make_pdu
        want_car = self.publish_carrier
        while True:
            r = self._rx.poll(cap=1024, psdu_stride=2048, want_csi=self.publish_csi, want_stats=self.snr_probe is not None,
                              trim_psdu=True)
            fr = r["frames"]
            nf = len(fr)
            if nf == 0:
                return
            flags = fr["flags"].tolist()
            n_out = fr["n_sym_out"].tolist()
            if self.snr_probe is not None:
                st = r["sym_stats"].tolist()
                for i in range(nf):
                    if n_out[i] > 0:
                        self.snr_probe.update_frame(st[i][0], st[i][1], st[i][2], 48 * n_out[i])
            if want_car and r["carrier"] is not None:
                car, p_car = r["carrier"], self._p_car
                if not grshim.HAVE_GNURADIO:
                    # on the shim a PDU is (dict, row view): rows by iteration over the frame's block (no index arithmetic per symbol)
                    for i in range(nf):
                        for row in car[i, :n_out[i]]:
                            pub(p_car, ({}, row))
                else:
                    for i in range(nf):
                        for sy in range(n_out[i]):
                            pub(p_car, make({}, car[i, sy]))
            ok = [i for i in range(nf) if flags[i] & capi.F_CRC_OK]
            self.frames_ok += len(ok)
            self.frames_dropped += nf - len(ok)
            if not ok:
                continue
            plen = fr["psdu_len"].tolist()
            enc = fr["encoding"].tolist()
            snr = fr["snr_db"].tolist()
            # sync_long's wifi_start tag as upstream forms it: (double)cfo_coarse - (double)cfo_fine
            foff = ((fr["cfo_coarse"].astype(np.float64) - fr["cfo_fine"].astype(np.float64))
                    * (self.bandwidth / (2 * math.pi))).tolist()
            psdu = r["psdu"]
            freq = self.frequency
            csi = r["csi"] if self.publish_csi else None
            p_mac = self._p_mac
            if csi is None and not grshim.HAVE_GNURADIO:
                # the common case on the shim, kept tight: a PDU is (dict, row view) -- poll() handed out its own copy of the rows
                for i in ok:
                    pub(p_mac, ({"frame_bytes": plen[i], "encoding": enc[i], "snr": snr[i], "freq": freq,
                                 "freq_offset": foff[i], "dlt": LINKTYPE_IEEE802_11}, psdu[i, :plen[i] - 4]))
                continue
            for i in ok:
                meta = {"frame_bytes": plen[i], "encoding": enc[i], "snr": snr[i], "freq": freq,
                        "freq_offset": foff[i], "dlt": LINKTYPE_IEEE802_11}
                if csi is not None:
                    meta["csi"] = csi[i]
                pub(p_mac, make(meta, psdu[i, :plen[i] - 4]))

    def get_probe_snr(self):
        """latest estimate of the SNR probe in dB (None without a probe)"""
        return None if self.snr_probe is None else self.snr_probe.snr()

    def stats(self):
        return self._rx.stats()

    def close(self):
        """Release the library handle (device buffers, stream)."""
        self._rx.close()
