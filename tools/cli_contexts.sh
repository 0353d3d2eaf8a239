#!/bin/bash
# tools/cli_contexts.sh: the C++ Renderer (host/rtr_renderer.h) on C2 with one context and with two contexts on GPU 0:
# last_seconds of the second render() call (scene already on the GPUs)
cd "$(dirname "$0")/.." || exit 1
cli=ray_tracing-rendering_amd/rtr_cli
for d in 0 0,0 0,0,0,0; do
  echo "== --devices $d"
  $cli 21 4 --width 800 --spp 400 --devices $d --repeat 3 2>&1 | grep "Rendering finished\|Msamples\|contexts"
done
