#!/bin/bash
# usage: tools/isa.sh <file.hip> <out.s>   - gfx950 device assembly of one translation unit (for reading the ISA)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -I../../include -S --cuda-device-only "$1" -o "$2"
