import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import modelgen as mg, parity
from oracle import mc_oracle as mo
import metalchat_amd as mc
import test_prefill_gpu as T
acc = mc.HardwareAccelerator()
for n in (65, 130, 300):
    for layers in (1, 2):
        cfg = mg.tiny_cfg(0, dim=1024, n_heads=8, n_kv_heads=2, head_dim=128, ffn_dim=2048, n_layers=layers, vocab=512, max_seq_len=320)
        for lora in (8, 0):
            weights = mg.make_model(cfg, seed=85, quant="i4", group=128, **({"lora_rank": lora} if lora else {}))
            tokens = np.random.default_rng(n).integers(0, cfg["vocab"], n).tolist()
            try:
                T.check_against_oracle(acc, cfg, weights, dict(weight_format=2, group_size=128), tokens, follow=1)
                print(os.environ.get("MC_PF3"), n, layers, lora, "ok", flush=True)
            except AssertionError as e:
                print(os.environ.get("MC_PF3"), n, layers, lora, "FAIL", str(e)[:120], flush=True)
