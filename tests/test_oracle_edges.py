"""Edge cases of the path through the oracle (what the reference would do with them is described in
SURVEY.md App. A; the GPU must then match the oracle on the very same batches: test_gpu_edges.py)."""
import numpy as np
import pytest

from cases import edge_cases

CASES = {c[0]: c for c in edge_cases()}


@pytest.mark.parametrize("name", sorted(CASES))
def test_edge_case_expectations(orc, name):
    _, iq, slot_len, max_sym, exp = CASES[name]
    prm = orc.make_params(max_sym=max_sym, llr_bits=6)
    o = orc.demod_batch(iq, slot_len, prm, want_eq=True)
    fr = o["frames"]
    psdu = orc.decode_batch(fr, o["idx"], prm, psdu_stride=2048)
    n_complete = int(((fr["flags"] & orc.F_COMPLETE) != 0).sum())
    if "complete" in exp:
        assert n_complete == exp["complete"]
    if "min_complete" in exp:
        assert n_complete >= exp["min_complete"]
    if "detected" in exp:
        assert int(((fr["flags"] & orc.F_DETECTED) != 0).sum()) == exp["detected"]
        assert (fr["trigger"][(fr["flags"] & orc.F_DETECTED) == 0] == -1).all()
    if "truncated" in exp:
        assert int(((fr["flags"] & 0x80) != 0).sum()) == exp["truncated"]
        assert (fr["n_sym_out"] < fr["n_sym"]).all()
    if "crc_ok" in exp:
        assert int(((fr["flags"] & orc.F_CRC_OK) != 0).sum()) == exp["crc_ok"]
    if "psdu" in exp:
        assert np.array_equal(psdu[:, :exp["psdu"].shape[1]], exp["psdu"])
    # incomplete frames never carry a PDU
    assert not ((fr["flags"] & orc.F_CRC_OK) & ~((fr["flags"] & orc.F_COMPLETE) << 3)).any()


def test_empty_batch(orc):
    prm = orc.make_params(max_sym=4)
    o = orc.demod_batch(np.zeros(0, np.complex64), 1024, prm)
    assert o["frames"].shape == (0,)


def test_stream_retrigger_rules(orc):
    """sync_short state machine: a second plateau inside MIN_GAP is ignored, later ones re-trigger."""
    from wifirx import txgen
    psdu = txgen.make_psdus(1, 40, seed=9)
    tx = txgen.encode_psdus(psdu, 0)
    s = tx.samples[0] * 6
    n = s.size
    x = np.zeros(6000, np.complex64)
    x[100:100 + n] += s
    x[100 + 200:100 + 200 + 160] += s[:160] * 3      # a strong short preamble 200 samples later: inside the gap
    x[3000:3000 + n] += s
    x += ((np.random.default_rng(0).standard_normal(x.size) + 1j * np.random.default_rng(1).standard_normal(x.size)) * 0.3).astype(np.complex64)
    trig, cfo = orc.sync_short(x)
    assert len(trig) == 2 and 100 < trig[0] < 260 and 3000 < trig[1] < 3160
    t1, _ = orc.sync_short(x, first_only=True)
    assert len(t1) == 1 and t1[0] == trig[0]
    # spec and libm detection agree on the trigger positions
    t2, _ = orc.sync_short(x, math_mode=orc.MATH_LIBM)
    assert np.array_equal(t2, trig)
