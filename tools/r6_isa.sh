#!/bin/bash
# Disassemble one translation unit's gfx950 code object into /tmp/isa/<stem>.s and cut one kernel (mangled symbol) into /tmp/isa/k.s. Development aid (round 6).
set -e
stem=${1:-octo_inst_p1}; sym=$2
mkdir -p /tmp/isa
L=/opt/rocm/lib/llvm/bin
$L/llvm-objcopy --dump-section=.hip_fatbin=/tmp/isa/$stem.fatbin /root/repo/octofitter.jl_amd/csrc/build/$stem.o
$L/clang-offload-bundler --unbundle --type=o --input=/tmp/isa/$stem.fatbin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=/tmp/isa/$stem.co
$L/llvm-objdump -d --no-show-raw-insn /tmp/isa/$stem.co > /tmp/isa/$stem.s
if [ -n "$sym" ]; then
  awk -v s="<$sym>:" 'index($0,s){p=1} p&&/^$/{if(n++>0)exit} p{print}' /tmp/isa/$stem.s | sed 's/ *\/\/ \([0-9A-F]*\):.*/  ;\1/' > /tmp/isa/k.s
  wc -l /tmp/isa/k.s
fi
