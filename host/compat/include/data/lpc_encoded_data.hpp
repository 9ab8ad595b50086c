// host/compat/include/data/lpc_encoded_data.hpp -- the reference's header path (src/include/data/lpc_encoded_data.hpp) over this host's classes:
// a program written against the reference's tree -- its own src/main.cpp -- compiles against this host with
//     g++ -I host/compat -I- -I host/include -I include ...
// (-I-: quoted includes are looked up in the -I directories instead of beside the including file).
#ifndef SELA_COMPAT_DATA_LPC_ENCODED_DATA_HPP
#define SELA_COMPAT_DATA_LPC_ENCODED_DATA_HPP
#include "sela_host/data.hpp"
#endif
