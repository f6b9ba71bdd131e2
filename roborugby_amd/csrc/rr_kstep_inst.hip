// rr_kstep_inst.hip -- explicit instantiations of the step kernel for one share (-DRR_PART=k, k < RR_KSTEP_PARTS) of the built
// configurations: the product build compiles the shares in parallel (roborugby_amd/build.py); see rr_kstep.hpp.
#include "rr_kstep.hpp"

#ifndef RR_PART
#error "compile with -DRR_PART=<0..RR_KSTEP_PARTS-1>"
#endif
#define RR_NOTHING
template <int PART> struct rr_part_tag {};
// (a macro cannot compare its argument with RR_PART, so every row expands to a constexpr-guarded nothing or to its instantiations
// through a second macro level keyed on the row's own part number)
#define RR_IF_PART_0(...)
#define RR_IF_PART_1(...)
#define RR_IF_PART_2(...)
#define RR_IF_PART_3(...)
#define RR_IF_PART_4(...)
#define RR_IF_PART_5(...)
#define RR_IF_PART_6(...)
#if RR_PART == 0
#undef RR_IF_PART_0
#define RR_IF_PART_0(...) __VA_ARGS__
#elif RR_PART == 1
#undef RR_IF_PART_1
#define RR_IF_PART_1(...) __VA_ARGS__
#elif RR_PART == 2
#undef RR_IF_PART_2
#define RR_IF_PART_2(...) __VA_ARGS__
#elif RR_PART == 3
#undef RR_IF_PART_3
#define RR_IF_PART_3(...) __VA_ARGS__
#elif RR_PART == 4
#undef RR_IF_PART_4
#define RR_IF_PART_4(...) __VA_ARGS__
#elif RR_PART == 5
#undef RR_IF_PART_5
#define RR_IF_PART_5(...) __VA_ARGS__
#elif RR_PART == 6
#undef RR_IF_PART_6
#define RR_IF_PART_6(...) __VA_ARGS__
#else
#error "RR_PART out of range"
#endif

#define X(part, a, b, c, d, R_, vw_, def_) \
    RR_IF_PART_##part(RR_KSTEP_VARIANTS(RR_NOTHING, a, b, c, d, R_, vw_, float, def_) RR_KSTEP_VARIANTS(RR_NOTHING, a, b, c, d, R_, vw_, double, 0))
RR_FOR_EACH_CFG_F64_PARTS(X)
#undef X
#define X(part, a, b, c, d, R_, vw_, def_) RR_IF_PART_##part(RR_KSTEP_VARIANTS(RR_NOTHING, a, b, c, d, R_, vw_, float, def_))
RR_FOR_EACH_CFG_F32_PARTS(X)
#undef X
#define X(part, a, b, c, d, R_, vw_, def_) RR_IF_PART_##part(RR_KSTEP_PLAIN(RR_NOTHING, a, b, c, d, R_, vw_, float) RR_KSTEP_PLAIN(RR_NOTHING, a, b, c, d, R_, vw_, double))
RR_FOR_EACH_CFG_F32S_PARTS(X)
#undef X
