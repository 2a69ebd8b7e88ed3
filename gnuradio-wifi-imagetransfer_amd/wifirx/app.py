"""Application layer of the reference's image demo around the PHY.

* piece wire format of upload_image_udp.py:19-34: ``struct.pack("=L", len) + pickle.dumps(((y,x,c), piece))``
* the ``Extract Pics`` block of gnu_radio/IRS_AP_epy_block_2.py:27-38: strip the 24-byte MAC header and
  the 4-byte length prefix, forward the pickle to UDP localhost:10010
* image rebuild of image_detach_rebuild.py:6-67 / download_image_udp.py:36-44

The reference's receiver calls ``pickle.loads`` on whatever arrives by UDP; :func:`load_piece` only
accepts the NumPy reconstruction opcodes a piece needs.
"""
from __future__ import annotations

import io
import pickle
import socket
import struct

import numpy as np

from . import grshim

PIECE_SIZE = (10, 10)


def detach_image_sorted(image: np.ndarray, piece_size=PIECE_SIZE):
    """``detach_image`` of the reference without the unseeded shuffle: pieces sorted by (y, x, c)."""
    h, w, ch = image.shape
    pieces = []
    for y in range(0, h, piece_size[1]):
        for x in range(0, w, piece_size[0]):
            for c in range(ch):
                pieces.append(((y, x, c), image[y:y + piece_size[1], x:x + piece_size[0], c:c + 1]))
    return pieces


def pack_piece(piece) -> bytes:
    data = pickle.dumps(piece)
    return struct.pack("=L", len(data)) + data


class _PieceUnpickler(pickle.Unpickler):
    _ALLOWED = {
        ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
        ("numpy", "ndarray"), ("numpy", "dtype"),
        ("numpy.core.numeric", "_frombuffer"), ("numpy._core.numeric", "_frombuffer"),
    }

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError("refusing %s.%s" % (module, name))


def load_piece(data: bytes):
    return _PieceUnpickler(io.BytesIO(data)).load()


def redraw_image(patch, image: np.ndarray, piece_size=PIECE_SIZE) -> np.ndarray:
    (y, x, c), piece = patch
    image[y:y + piece_size[1], x:x + piece_size[0], c:c + 1] = piece
    return image


class extract_pics(grshim.basic_block):
    """Message sink with port ``MAC``: same slicing as the reference's Extract Pics block; the payload
    goes to ``sink(bytes)`` (default: UDP datagram to localhost:10010)."""

    def __init__(self, sink=None, host="localhost", port=10010):
        grshim.basic_block.__init__(self, name="Extract Pics", in_sig=None, out_sig=None)
        self.message_port_register_in(grshim.intern("MAC"))
        self.set_msg_handler(grshim.intern("MAC"), self.handle_msg)
        self._sink = sink
        self._addr = (host, port)
        self._skt = None if sink is not None else socket.socket(socket.AF_INET, socket.SOCK_DGRAM)

    def handle_msg(self, msg):
        data = grshim.to_python(msg)[-1]
        target = data[24:]
        pics = bytes(target[4:])
        if self._sink is not None:
            self._sink(pics)
        else:
            self._skt.sendto(pics, self._addr)
