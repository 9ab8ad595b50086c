// host/compat/include/rice.hpp -- the reference's src/include/rice.hpp path.  The Rice coder of this host lives on the device, inside the
// frame kernels (sela_amd/csrc/): there is no per-stage host class behind this header; src/main.cpp includes it and uses
// nothing of it.
#ifndef SELA_COMPAT_RICE_HPP
#define SELA_COMPAT_RICE_HPP
#include "sela_host/data.hpp"
#endif
