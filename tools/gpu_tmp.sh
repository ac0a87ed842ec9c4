python -m pytest tests/test_round2_gpu.py -m gpu -q --timeout 600 -k "registers or default_small" 2>&1 | tail -3
python tools/shape_scan.py 512 1024 1536 2048 --quick 2>&1 | grep -E "== N|auto|direct|fused_ipl2_ls64_tl4" | cut -c1-120
