#!/bin/bash
# gpurun call 33 of round 2: mel-VAE encoder on the engine + the refactored decoder attention block
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2; mkdir -p $O
timeout 500 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -s -k "vae_encoder or vae_and_vocoder or vae_vocoder_match" > $O/vae_enc.log 2>&1; echo "rc=$?"; tail -3 $O/vae_enc.log; grep -E "rel err|Error|error" $O/vae_enc.log | head
