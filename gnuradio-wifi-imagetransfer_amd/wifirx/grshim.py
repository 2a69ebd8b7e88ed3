"""GNU-Radio-free stand-in for the few pieces of ``gnuradio.gr`` / ``pmt`` the reference's Python
blocks use (gnu_radio/IRS_AP_epy_block_2.py:11-22): ``sync_block`` with message ports and
``work()``.  GNU Radio is absent in this image and on the GPU box, so the replacement block runs
on this shim; when ``gnuradio`` is importable the real classes are used instead and the same
block drops into a flowgraph.

A PDU on the shim is the Python tuple ``(dict, numpy.ndarray)`` -- exactly what
``pmt.to_python(msg)`` hands the reference's consumers (``pmt.to_python(msg)[-1]`` at
gnu_radio/IRS_AP_epy_block_2.py:31, ``[0]['snr']`` at gnu_radio/IRS_tranceiver.grc:924-925).
"""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - GNU Radio is not installed in the build image
    from gnuradio import gr as _gr
    import pmt as _pmt
    HAVE_GNURADIO = True
except Exception:  # ImportError or a broken install
    _gr = None
    _pmt = None
    HAVE_GNURADIO = False


def intern(name: str):
    return _pmt.intern(name) if HAVE_GNURADIO else name


def make_pdu(meta: dict, vec: np.ndarray):
    """``pmt.cons(dict, uniform vector)`` under GNU Radio, ``(dict, ndarray)`` on the shim."""
    if HAVE_GNURADIO:  # pragma: no cover
        d = _pmt.to_pmt(dict(meta))
        return _pmt.cons(d, _pmt.to_pmt(np.ascontiguousarray(vec)))
    return (meta, vec if vec.flags.c_contiguous else np.ascontiguousarray(vec))


def to_python(msg):
    return _pmt.to_python(msg) if HAVE_GNURADIO else msg


if HAVE_GNURADIO:  # pragma: no cover
    sync_block = _gr.sync_block
    basic_block = _gr.basic_block
else:
    class basic_block:
        """Message-port half of gr.basic_block."""

        def __init__(self, name="", in_sig=None, out_sig=None):
            self._name = name
            self.in_sig = in_sig
            self.out_sig = out_sig
            self._out_ports = {}
            self._handlers = {}

        def name(self):
            return self._name

        def message_port_register_out(self, port):
            self._out_ports.setdefault(port, [])

        def message_port_register_in(self, port):
            self._handlers.setdefault(port, None)

        def set_msg_handler(self, port, fn):
            self._handlers[port] = fn

        def message_port_pub(self, port, msg):
            for dst, dport in self._out_ports[port]:
                h = dst._handlers.get(dport)
                if h is not None:
                    h(msg)

    class sync_block(basic_block):
        """Stream half: the scheduler is replaced by :func:`run_stream`."""

        def work(self, input_items, output_items):  # pragma: no cover - overridden
            raise NotImplementedError


class sink_block(basic_block if not HAVE_GNURADIO else object):
    """Message sink with one input port ``in`` that hands every PDU (as Python) to ``fn`` -- what
    ``blocks.message_debug`` is to a flowgraph, for the shim's harnesses (bench host_path leg, tests)."""

    def __init__(self, fn):
        basic_block.__init__(self, name="sink", in_sig=None, out_sig=None)
        self.message_port_register_in(intern("in"))
        self.set_msg_handler(intern("in"), lambda msg: fn(to_python(msg)))


class pmt_shim:
    """The handful of ``pmt`` functions the reference's Python blocks and INTEGRATION.md's stub call, over plain Python
    values -- ``pmt.to_python`` of the real thing yields the same objects.  tests/test_integration_stub.py installs it
    (and :class:`gr_shim`) as ``sys.modules['pmt']`` / ``['gnuradio']`` to execute the documented binding."""
    PMT_NIL = None
    intern = staticmethod(lambda name: name)
    cons = staticmethod(lambda a, b: (a, b))
    make_dict = staticmethod(dict)
    to_pmt = staticmethod(lambda v: v.copy() if isinstance(v, np.ndarray) else v)     # the real one copies into a PMT vector
    to_python = staticmethod(lambda v: v)
    car = staticmethod(lambda p: p[0])
    cdr = staticmethod(lambda p: p[1])
    init_u8vector = staticmethod(lambda n, items: np.asarray(items, dtype=np.uint8)[:n])
    u8vector_elements = staticmethod(lambda v: list(np.asarray(v, dtype=np.uint8)))


class gr_shim:
    """``from gnuradio import gr`` for the shim: the two block classes."""
    sync_block = sync_block
    basic_block = basic_block


def msg_connect(src, sport, dst, dport):
    """``tb.msg_connect((src, sport), (dst, dport))`` for shim blocks."""
    src._out_ports[sport].append((dst, dport))


def run_stream(block, samples: np.ndarray, chunk: int = 8192, finish: bool = True) -> int:
    """Drive ``block.work()`` the way the GNU Radio scheduler would: successive chunks of the input
    stream, each call consuming what ``work`` returns."""
    samples = np.ascontiguousarray(samples, dtype=np.complex64)
    pos = stalls = 0
    while pos < samples.size:
        n = block.work([samples[pos:pos + chunk]], [])
        if n == -1:                     # WORK_DONE: the block has ended the flowgraph (a dead stream, wifirx/block.py)
            break
        if n <= 0:                      # nothing consumed: the scheduler calls again; give up after a few in a row
            stalls += 1
            if stalls > 8:
                break
            continue
        stalls = 0
        pos += n
    if finish and hasattr(block, "stop"):
        block.stop()
    return pos
