// beom_hash.cpp — sha1 prefix of the sources libbeom_hip.so was built from (Makefile), compared with
// the tree by the bindings (beom_amd/capi.py) so that a stale library is refused.
#ifndef BEOM_SRC_HASH
#define BEOM_SRC_HASH "unknown"
#endif
extern "C" const char *beom_source_hash(void) { return BEOM_SRC_HASH; }
