// g++ -Iscratch/hoststub -Islimfastq_amd/csrc scratch/host_chain_test.cpp: LaneEncB against LaneEnc on random symbol streams (one "lane")
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "dev_chain.h"
int main() {
    srand(7);
    for (int trial = 0; trial < 20000; trial++) {
        const int nsym = rand() % 3000;
        const int mode = rand() % 3;
        std::vector<uint8_t> o1(nsym * 4 + 64, 0xAA), o2(nsym * 4 + 64, 0xBB);
        const u32 cap = (u32)((nsym * 4 + 32) & ~15);
        LaneEnc a; a.init(o1.data(), cap);
        static u32 ring[LaneEncB<1, 8>::LDS_DWORDS];
        LaneEncB<1, 8> b; b.init(ring, 0, o2.data(), cap);
        int since = 0;
        for (int i = 0; i < nsym; i++) {
            if (rand() % 97 == 0) {          // force the carry-less corner (coder.hpp:76-77): bits 24..55 of low all ones
                const u64 forced = ((u64)(rand() & 0xff) << 56) | 0x00FFFFFFFF000000ull | (u64)(rand() & 0xFFFFFF);
                a.low = b.low = forced;
            }
            const bool valid = rand() % 8 != 0;
            u32 cum, freq, tot;
            if (mode == 0) { tot = 65536; freq = 1 + rand() % (rand() % 4 ? 60000 : 3); cum = rand() % (tot - freq + 1); }
            else if (mode == 1) { tot = 12; freq = 3; cum = 3 * (rand() % 4); }
            else { tot = 4 + rand() % 1017; freq = 1 + rand() % (tot < 256 ? tot : 255); if (freq > tot) freq = tot; cum = rand() % (tot - freq + 1); }
            if (mode == 0) { if (valid) a.encode16(cum, freq); b.encode16_if(valid ? ~0u : 0u, cum, freq); }
            else { const u32 rc = fz_recip(tot); if (valid) a.encode(cum, freq, tot, rc); b.encode_if(valid ? ~0u : 0u, cum, freq, tot, rc); }
            if (++since == 4) { b.drain(); since = 0; }
        }
        const u32 s1 = a.finish(), s2 = b.finish();
        if (s1 != s2 || a.err != b.err || memcmp(o1.data(), o2.data(), s1)) { printf("MISMATCH trial %d nsym %d mode %d sizes %u %u err %u %u\n", trial, nsym, mode, s1, s2, a.err, b.err); return 1; }
        for (size_t k = cap; k < o2.size(); k++) if (o2[k] != 0xBB) { printf("write past cap trial %d\n", trial); return 1; }
    }
    printf("ok\n");
    return 0;
}
