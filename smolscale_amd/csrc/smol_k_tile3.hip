/* k_tile instantiations for tap reader 3 (see smol_k_tile.inc.h) */
#define SMOL_TILE_TS 3
#include "smol_k_tile.inc.h"
