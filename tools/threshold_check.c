// Exhaustive check that int32(v * q) of the reference (src/pile.cpp:94, IEEE double) equals the integer
// expressions the run-space kernel uses, for every 16-bit coverage value: q = 1.3 -> v * 13 / 10,
// 1.82 -> v * 182 / 100, 1.42 -> v * 142 / 100.  gcc -O2 tools/threshold_check.c && ./a.out  ->  "0 0 0"
#include <stdio.h>
#include <stdint.h>
int main(){ long bad13=0,bad182=0,bad142=0; for (uint32_t v=0; v<=65535; ++v){ int32_t a=(int32_t)((double)v*1.3); int32_t b=(int32_t)((uint64_t)v*13/10); if(a!=b) ++bad13; a=(int32_t)((double)v*1.82); b=(int32_t)((uint64_t)v*182/100); if(a!=b) ++bad182; a=(int32_t)((double)v*1.42); b=(int32_t)((uint64_t)v*142/100); if(a!=b) ++bad142;} printf("%ld %ld %ld\n",bad13,bad182,bad142); return 0; }
