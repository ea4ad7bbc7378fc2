#!/bin/bash
# tools/isa_of.sh <mangled-name-substring> [source stem]: print the gfx950 assembly of ONE kernel of the product build (build/obj)
stem=${2:-mfx_sgd_res}
awk -v pat="$1" 'index($0, "_Z") == 1 && index($0, pat) && index($0, ": ") {f=1} f{print} /s_endpgm/{if(f){exit}}' /root/repo/build/obj/$stem-hip-amdgcn-amd-amdhsa-gfx950.s
