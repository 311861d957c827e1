// Host-side check of the IVF-PQ code layouts (rsx_internal.h: pq_code_addr) — built and run by tests/test_layout_host.py (no GPU needed).
//  * every layout maps the (row, m) pairs of a 64-vector slab one-to-one onto the slab's 64 * Mpad bytes;
//  * the rotated layouts hand a scan lane exactly the bytes k_pq_scan_rot's gather addresses assume: lane (g, i) of a 16-vector block
//    reads 16 contiguous bytes at p * 1024 + lane * 16 whose byte s is sub-quantiser 64 p + 16 g + ((i + s) & 15) of vector i (and the
//    8-byte half phase likewise); for M = 16 lane l of a 64-vector block reads vector l's 16 bytes, byte s = sub-quantiser (l + s) & 15;
//  * at every step the 32 lanes of a half wave address 32 different LDS banks of the [code][m] table (bank = m % 32; M = 16: the
//    duplicated row, copy g & 1).
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>
#include "../../retrieval-scaling_amd/csrc/rsx_internal.h"

using namespace rsx;

static int fail(const char* what, int M, int a, int b) { std::printf("FAIL %s M=%d (%d, %d)\n", what, M, a, b); return 1; }

int main() {
    // 1. bijection per slab, all layouts the engine can choose
    struct L { int M, Mpad, CB; };
    const L layouts[] = {{8, 16, 16}, {12, 16, 16}, {16, 16, 16}, {16, 16, 0}, {32, 32, 0}, {32, 32, 16}, {48, 48, 16}, {64, 64, 0}, {96, 96, 0}, {96, 96, 16},
                         {128, 128, 0}, {160, 160, 4}, {20, 32, 4}, {96, 96, PQ_SLICED}, {64, 64, PQ_SLICED}, {32, 32, PQ_SLICED}, {128, 128, PQ_SLICED}};
    for (const L& l : layouts) {
        const int unit = l.CB == PQ_SLICED ? 32 * PQ_SLICED_GB : 64;       // the sliced layout is slice-major over groups of 16 blocks = 512 vectors
        for (int64_t slab = 0; slab < 3; slab++) {
            std::set<int64_t> seen;
            for (int v = 0; v < unit; v++)
                for (int m = 0; m < l.Mpad; m++) {
                    const int64_t a = pq_code_addr(slab * unit + v, m, l.Mpad, l.CB);
                    if (a < slab * unit * l.Mpad || a >= (slab + 1) * unit * l.Mpad) return fail("address outside the slab / group", l.M, v, m);
                    if (!seen.insert(a).second) return fail("two (row, m) share a byte", l.M, v, m);
                }
        }
    }
    // 2. rotated layouts: what a scan lane reads
    for (int M : {32, 64, 96, 128}) {
        const int NF = M >> 6, NH = (M >> 5) & 1;
        for (int blk = 0; blk < 5; blk++)
            for (int lane = 0; lane < 64; lane++) {
                const int g = lane >> 4, i = lane & 15;
                for (int p = 0; p < NF; p++)
                    for (int s = 0; s < 16; s++) {
                        const int m = 64 * p + 16 * g + ((i + s) & 15);
                        if (pq_code_addr(blk * 16 + i, m, M, 0) != (int64_t)blk * 16 * M + p * 1024 + lane * 16 + s) return fail("full phase byte", M, lane, s);
                    }
                if (NH)
                    for (int s = 0; s < 8; s++) {
                        const int m = 64 * NF + 16 * (g & 1) + ((i + s + 8 * (g >> 1)) & 15);
                        if (pq_code_addr(blk * 16 + i, m, M, 0) != (int64_t)blk * 16 * M + NF * 1024 + lane * 8 + s) return fail("half phase byte", M, lane, s);
                    }
            }
        // banks: step s of a phase, half wave h
        for (int s = 0; s < 16; s++)
            for (int h = 0; h < 2; h++) {
                std::set<int> banks;
                for (int lane = 32 * h; lane < 32 * h + 32; lane++) { const int g = lane >> 4, i = lane & 15; banks.insert((16 * g + ((i + s) & 15)) % 32); }
                if (NF && banks.size() != 32) return fail("full phase banks", M, s, h);
            }
        if (NH)
            for (int s = 0; s < 8; s++)
                for (int h = 0; h < 2; h++) {
                    std::set<int> banks;
                    for (int lane = 32 * h; lane < 32 * h + 32; lane++) { const int g = lane >> 4, i = lane & 15; banks.insert((16 * (g & 1) + ((i + s + 8 * (g >> 1)) & 15)) % 32); }
                    if (banks.size() != 32) return fail("half phase banks", M, s, h);
                }
    }
    {   // M = 16: 64-vector blocks
        for (int blk = 0; blk < 5; blk++)
            for (int lane = 0; lane < 64; lane++)
                for (int s = 0; s < 16; s++)
                    if (pq_code_addr(blk * 64 + lane, (lane + s) & 15, 16, 0) != (int64_t)blk * 1024 + lane * 16 + s) return fail("M = 16 byte", 16, lane, s);
        for (int s = 0; s < 16; s++)
            for (int h = 0; h < 2; h++) {
                std::set<int> banks;     // table row: 64 dwords, entry m at dword m (copy 0) and 16 + m (copy 1); lane group g uses copy g & 1
                for (int lane = 32 * h; lane < 32 * h + 32; lane++) { const int g = lane >> 4, i = lane & 15; banks.insert((16 * (g & 1) + ((i + s) & 15)) % 32); }
                if (banks.size() != 32) return fail("M = 16 banks", 16, s, h);
            }
    }
    // 3. sliced layout (PQ_SLICED): lane (g, i) of a 32-vector block reads, per slice, 16 contiguous bytes at slice * 1024 + lane * 16 whose
    // byte b is sub-quantiser 32 slice + 16 (g & 1) + ((i + b) & 15) of vector 16 (g >> 1) + i; the 8-byte-entry table of a slice is
    // [code][m & 31] x 8 B (256-byte rows): ds_read_b64 banks = (addr / 4) % 64, so a half wave must address 32 different 8-byte slots;
    // the four-query pre-pass image is [code][m & 31] x 4 B per row half: bank = m % 32
    for (int M : {32, 64, 96, 128}) {
        // the same slice of the 16 blocks of a group is ONE contiguous 16 KiB run (what the 16 waves of a scan workgroup read at a time)
        for (int grp = 0; grp < 3; grp++)
            for (int sl = 0; sl < M / 32; sl++)
                for (int bi = 0; bi < PQ_SLICED_GB; bi++)
                    if (pq_sliced_off(grp * PQ_SLICED_GB + bi, sl, M) != (int64_t)grp * PQ_SLICED_GB * 32 * M + sl * PQ_SLICED_GB * 1024 + bi * 1024) return fail("group run", M, sl, bi);
        for (int blk = 0; blk < 37; blk++)
            for (int lane = 0; lane < 64; lane++) {
                const int g = lane >> 4, i = lane & 15;
                for (int sl = 0; sl < M / 32; sl++)
                    for (int b = 0; b < 16; b++) {
                        const int m = 32 * sl + 16 * (g & 1) + ((i + b) & 15);
                        if (pq_code_addr(blk * 32 + 16 * (g >> 1) + i, m, M, PQ_SLICED) != pq_sliced_off(blk, sl, M) + lane * 16 + b) return fail("sliced byte", M, lane, b);
                    }
            }
        for (int s = 0; s < 16; s++)
            for (int h = 0; h < 2; h++) {
                std::set<int> slots;
                for (int lane = 32 * h; lane < 32 * h + 32; lane++) { const int g = lane >> 4, i = lane & 15; slots.insert(16 * (g & 1) + ((i + s) & 15)); }
                if (slots.size() != 32) return fail("sliced banks", M, s, h);
            }
    }
    std::printf("layouts ok\n");
    return 0;
}
