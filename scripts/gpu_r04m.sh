#!/bin/bash
# Round 4, final evidence, part 1: PMC passes -> traffic.json, the default bench line, kernel trace of the same command (stage b of
# scripts/gpu_round_end.sh), then the end-to-end shim (stage c).
TAG=${TAG:-r04m} STAGES=bc bash scripts/gpu_round_end.sh
