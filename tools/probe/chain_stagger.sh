for s in 0 8 16 32 64; do echo "stagger $s: $(FDET_CHAIN_STAGGER=$s timeout -k 10 120 python tools/probe/chain_time.py 2>&1 | tail -1)"; done
