#!/usr/bin/env python3
"""Tuning aid: clocks and socket power while one GEMV variant is launched back to back.
usage: power_probe.py <mode>   mode: m4d | m4 | stream | fast"""
import os, subprocess, sys, threading, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mode = sys.argv[1]
os.environ["MC_GEMV_M4"] = {"m4d": "3", "m4": "1"}.get(mode, "2")
if mode == "stream":
    os.environ["MC_GEMV_DBG"] = "1"
import metalchat_amd as mc

acc = mc.HardwareAccelerator()
dec = mc.Decoder(acc, dtype=mc.BF16, dim=4096, n_heads=32, n_kv_heads=8, head_dim=128, ffn_dim=14336, n_layers=32,
                 vocab=128256, max_seq_len=64, rope_theta=500000.0, norm_eps=1e-5, attn_scale=128 ** -0.5,
                 weight_format=mc.WFMT_I4, group_size=128, qmode=mc.QMODE_FAST if mode == "fast" else mc.QMODE_EXACT)
dec.init_synthetic(1)
samples = []
stop = False


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout
            d = json.loads(out)
            c = d.get("card0", d[next(iter(d))])
            samples.append({k: v for k, v in c.items() if any(t in k.lower() for t in ("sclk", "mclk", "fclk", "socclk", "power"))})
        except Exception as e:  # noqa: BLE001
            samples.append({"error": str(e)[:80]})
        time.sleep(0.4)


th = threading.Thread(target=sampler)
th.start()
t0 = time.time()
per = []
while time.time() - t0 < 6.0:
    ms, by, ln = dec.time_gemv("w13", 50)
    per.append(ms / (50 * ln) * 1e3)
stop = True
th.join()
print(json.dumps(dict(mode=mode, us_first=round(per[0], 2), us_last=round(per[-1], 2), n=len(per))))
for s in samples[2:12]:
    print(json.dumps(s))
