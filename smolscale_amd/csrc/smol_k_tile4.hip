/* k_tile instantiations for tap reader 4 (see smol_k_tile.inc.h) */
#define SMOL_TILE_TS 4
#include "smol_k_tile.inc.h"
