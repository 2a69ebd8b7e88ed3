import os, sys, time
sys.path.insert(0, "gnuradio-wifi-imagetransfer_amd"); sys.path.insert(0, "tools")
import numpy as np
from host_path_bench import make_stream
from wifirx import block, grshim
x, n_sym = make_stream(2, 8000, 4608, 160)
blk = block.wifi_phy_rx(bandwidth=20e6, frequency=5.89e9, max_sym=n_sym, publish_carrier=False, batch_samples=1 << 22)
got = []
grshim.msg_connect(blk, "mac_out", grshim.sink_block(got.append), "in")
grshim.run_stream(blk, x, chunk=32768)
t = time.perf_counter(); grshim.run_stream(blk, x, chunk=32768); dt = time.perf_counter() - t
print("total", x.size / dt / 1e9, "Gsample/s", len(got), file=sys.stderr)
# pure ingest: time work() alone with publishing disabled
blk2 = block.wifi_phy_rx(bandwidth=20e6, frequency=5.89e9, max_sym=n_sym, publish_carrier=False, batch_samples=1 << 22)
blk2._publish = lambda: None
grshim.run_stream(blk2, x[:1<<22], chunk=32768, finish=False)
t = time.perf_counter(); grshim.run_stream(blk2, x, chunk=32768, finish=False); dt = time.perf_counter() - t
print("ingest only (no poll)", x.size / dt / 1e9, "Gsample/s", file=sys.stderr)
t = time.perf_counter(); grshim.run_stream(blk2, x, chunk=8192, finish=False); dt = time.perf_counter() - t
print("ingest only (no poll) 8192", x.size / dt / 1e9, "Gsample/s", file=sys.stderr)
