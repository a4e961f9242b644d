// types.hpp -- precision switch of the host mirror (reference: ver7/types.hpp:21).
// The reference edits the typedef by hand; here -DNBX_REAL_DOUBLE selects the fp64 build
// (BASELINE.json configs[4]) and the default stays float.
#ifndef NBX_HOST_TYPES_HPP
#define NBX_HOST_TYPES_HPP

#ifdef NBX_REAL_DOUBLE
typedef double real_type;
#else
typedef float real_type;
#endif

#endif
