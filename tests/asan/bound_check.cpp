// CPU-only sanitizer harness (g++ -fsanitize=address,undefined): calls the HIP-free shape checks of
// the C ABI (raiko_amd/csrc/taps.hpp: rk::check_taps, rk::seal_bound_words -- what
// rk_seal_bound_words, rk_prove_segment and rk_verify_segment_ex run first) on malformed tap sets
// held in exactly-sized heap arrays, so any out-of-bounds read aborts.
#include <cstdio>
#include <cstring>
#include <vector>

#include "taps.hpp"

struct Shape {
    std::vector<uint32_t> rg, ro, rc, off, backs;
    rk_segment seg;
    Shape(uint32_t wa, uint32_t wc, uint32_t wd) {
        std::memset(&seg, 0, sizeof seg);
        uint32_t gs[3] = {wa, wc, wd};
        for (uint32_t g = 0; g < 3; g++)
            for (uint32_t o = 0; o < gs[g]; o++) {
                rg.push_back(g);
                ro.push_back(o);
                rc.push_back(g == 0 ? 1 : 0);
            }
        off = {0, 1, 3};
        backs = {0, 0, 1};
        sync();
        seg.po2 = 10;
        for (int g = 0; g < 3; g++) seg.taps.group_size[g] = gs[g];
        seg.taps.n_combos = 2;
        seg.n_globals = 4;
    }
    void sync() {
        seg.taps.n_regs = (uint32_t)rg.size();
        seg.taps.reg_group = rg.data();
        seg.taps.reg_offset = ro.data();
        seg.taps.reg_combo = rc.data();
        seg.taps.combo_off = off.data();
        seg.taps.combo_backs = backs.data();
    }
};

static int fails = 0;
static void expect(bool ok, const char* what) {
    if (!ok) {
        std::printf("FAIL: %s\n", what);
        fails++;
    }
}

int main() {
    {
        Shape s(4, 4, 8);
        expect(rk::check_taps(s.seg.taps) == RK_OK, "well-formed taps accepted");
        expect(rk::seal_bound_words(&s.seg) > 0, "bound of a well-formed shape");
    }
    {
        Shape s(4, 4, 8);
        s.rc[0] = 99;  // combo id far outside combo_off
        expect(rk::check_taps(s.seg.taps) == RK_ERR_INVALID, "combo id out of range");
        expect(rk::seal_bound_words(&s.seg) == 0, "bound is 0 for a bad combo id");
    }
    {
        Shape s(4, 4, 8);
        s.off = {0, 3, 1};  // not monotone: differences would underflow
        s.sync();
        expect(rk::seal_bound_words(&s.seg) == 0, "non-monotone combo_off");
    }
    {
        Shape s(4, 4, 8);
        s.ro[5] = 77;  // offset outside its group
        expect(rk::seal_bound_words(&s.seg) == 0, "register offset out of range");
    }
    {
        Shape s(4, 4, 8);
        s.seg.taps.group_size[2] = 9;  // sizes do not add up to n_regs
        expect(rk::seal_bound_words(&s.seg) == 0, "group sizes vs n_regs");
    }
    {
        Shape s(4, 4, 8);
        s.backs = {0, 0, 200};
        s.sync();
        expect(rk::seal_bound_words(&s.seg) == 0, "back beyond the supported range");
    }
    {
        Shape s(4, 4, 8);
        s.seg.taps.combo_off = nullptr;
        expect(rk::seal_bound_words(&s.seg) == 0, "null array");
        s.sync();
        s.seg.po2 = 0;
        expect(rk::seal_bound_words(&s.seg) == 0, "po2 = 0");
        s.seg.po2 = 23;
        expect(rk::seal_bound_words(&s.seg) == 0, "po2 too large");
        expect(rk::seal_bound_words(nullptr) == 0, "null segment");
    }
    std::printf(fails ? "%d failures\n" : "ok\n", fails);
    return fails ? 1 : 0;
}
