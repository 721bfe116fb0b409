"""Sample gfx clock / socket power every 50 ms for N seconds (amdsmi); prints one line per sample.  usage: smi_loop.py <seconds>"""
import sys
import time

import amdsmi

amdsmi.amdsmi_init()
h = amdsmi.amdsmi_get_processor_handles()[0]
t0 = time.time()
while time.time() - t0 < float(sys.argv[1]):
    d = amdsmi.amdsmi_get_gpu_metrics_info(h)
    xs = [x for x in d.get("current_gfxclks", []) if isinstance(x, int) and 0 < x < 60000]
    print(f"{time.time() - t0:7.2f} s  gfxclk {sum(xs) / max(1, len(xs)):7.1f} MHz  power {d.get('current_socket_power')} W  hotspot {d.get('temperature_hotspot')} C", flush=True)
    time.sleep(0.05)
