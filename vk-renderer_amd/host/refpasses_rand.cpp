// refpasses_rand.cpp — only linked into build/refpasses/libvkr_host_refpasses.so (Makefile: refpasses, -Wl,--wrap=rand).
// The reference's GTAO::add_main_pass adds rand() / float(RAND_MAX) - 0.5 to the slice angle (gtao.cpp:111); the drop-in
// test compares against the oracle at a pinned angle, so rand() is made to return the value for which that term is
// exactly 0: float(RAND_MAX / 2) rounds to 2^30, float(RAND_MAX) to 2^31, their quotient is 0.5.
#include <cstdlib>
extern "C" int __wrap_rand(void) { return RAND_MAX / 2; }
