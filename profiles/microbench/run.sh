#!/bin/bash
# runs every microbenchmark entry in its own process under a timeout (an entry that hangs does not take the rest with it)
cd "$(dirname "$0")"
out=${1:-../../gpurun_out/mb/valu_lds_peak.txt}
mkdir -p "$(dirname "$out")"
: > "$out"
for n in add xor and or sub subrev shl shlv lshr ashr not mov min max addlit andlit adds addco addc mul24 bitop3 bitop3_maj bfi \
         perm alignbit alignbyte sdwa_shl sdwa_mul sdwa_sub sad lshl_add lshl_or and_or xad or3 add3 bfe cmp cmps cnds cndmask cmp_cnd \
         max3 dpp mad24 mul_lo bcnt mbcnt pkadd dot4 lshl64 add_bitop add_perm salu valu_salu readlane fma \
         "~ds_read_b32 lane*4" "~ds_read_b32 lane*152" "~ds_read_b32 lane*150" "~ds_read_b32 same" "~ds_read_b64 lane*8" "~ds_read_b64 lane*152" \
         "~ds_read_b128 lane*16" "~ds_read_b128 lane*152" "~ds_read_b128 lane*160" "~ds_read_b128 lane*144" "~ds_read_u8" \
         "~ds_add_u32 lane*4" "~ds_add_u32 lane*8" "~ds_add_u32 lane*128" "~ds_add_u32 same" "~ds_add_rtn" "~ds_write_b32" "~ds_bpermute"; do
    timeout 40 ./valu_lds_peak "$n" 2>&1 | grep -v "^#\|^instruction" >> "$out" || echo "$n: rc=$? (timeout or crash)" >> "$out"
done
