#!/bin/bash
# attention launch time around the L = 257 boundary (9 waves, 3/2/2/2 per SIMD) vs L = 256 (8 waves, 2 per SIMD, DEEP variant)
cd /root/repo
for L in 224 256 257 288; do L=$L python tools/attn_probe.py; done
OVHIP_ATTN_LONEKEY=0 L=257 python tools/attn_probe.py
OVHIP_ATTN_MODE=1 L=257 python tools/attn_probe.py
OVHIP_ATTN_MODE=2 L=257 python tools/attn_probe.py
OVHIP_ATTN_MODE=1 L=256 python tools/attn_probe.py
OVHIP_ATTN_MODE=2 L=256 python tools/attn_probe.py
