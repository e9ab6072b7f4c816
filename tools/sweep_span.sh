for ms in 256 512 1024 2048 4096; do echo "--- min_span $ms"; OCTO_SMALL_MIN_SPAN=$ms python tools/latency_vs_w.py 1,32,128,256,512 1024 2>&1 | grep "^E=" | grep -v "E=    50"; done
