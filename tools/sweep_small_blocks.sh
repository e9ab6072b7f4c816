for b in 32 64 128 256 512 1024; do echo "== OCTO_SMALL_BLOCKS=$b"; OCTO_SMALL_BLOCKS=$b python tools/latency_w1.py 2>&1 | grep "octo_eval E=10000.*small-batch"; done
