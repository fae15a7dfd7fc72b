for nt in "4 64 64 192 640 7" "4 256 256 24 80 3" "4 64 64 96 320 3"; do
for d in 0 1 2 8 10 11 15; do echo "shape $nt variant 0 dbg $d: $(PNSFM_AUTOTUNE=0 PNSFM_CONV_VARIANT=0 PNSFM_CONV_DBG=$d python tools/conv_micro.py $nt 20 fwd 2>&1 | tail -1)"; done; done
