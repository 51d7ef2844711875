#!/bin/bash
for ab in 0 10 74 202 207 2 8 128; do echo "== ablate $ab"; HGYM_ENV_ABLATE=$ab python tools/probe_env.py 2>&1 | grep "env_step_synth N=4096"; done
