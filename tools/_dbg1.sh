cd /root/repo
for v in 1 0; do echo "=== PNSFM_CAPTURE_SCRATCH=$v"; PNSFM_CAPTURE_SCRATCH=$v timeout 600 python tools/graph_repro.py 10 sgd 2>&1 | grep -E "^[0-9]+ graph|worst|Error|error" | cut -c1-300; done
