#!/bin/bash
# ISA of one kernel of a csrc/*.hip file:  tools/isa_dump.sh <file.hip> <kernel name regex> [extra hipcc flags]  -> /tmp/isa.s
f=$(realpath $1); k=$2; shift 2
inc=$(dirname $f)
d=$(mktemp -d)
( cd $d && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics "$@" -I$inc -c $f -o x.o -save-temps=obj >/dev/null 2>&1 )
s=$(ls $d/*gfx950*.s 2>/dev/null | head -1)
[ -z "$s" ] && { echo "compile failed"; rm -rf $d; exit 1; }
awk "/$k/,/s_endpgm/" $s > /tmp/isa.s
grep -E "vgpr_count|sgpr_count|group_segment_fixed_size|spill" $s | tail -40 > /tmp/isa_meta.txt
rm -rf $d
wc -l /tmp/isa.s
