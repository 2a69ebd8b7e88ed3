"""wifirx -- MI355X-native IEEE 802.11a/g OFDM receive chain behind the reference's block interface.

The compute path is libwifirx.so (hand-written HIP for gfx950) reached through the C ABI of
include/wifirx.h; this package holds the ctypes binding (capi), the GNU-Radio-shaped block that
mirrors the RX half of the reference's wifi_phy_hier (block, grshim), the CPU transmitter used as
synthetic source (txgen) and the application-layer helpers of the image demo (app).
"""
__all__ = ["capi", "txgen", "grshim", "block", "app", "dist"]
