/* k_tile instantiations for tap reader 0 (see smol_k_tile.inc.h) */
#define SMOL_TILE_TS 0
#include "smol_k_tile.inc.h"
