/* k_tile instantiations for tap reader 2 (see smol_k_tile.inc.h) */
#define SMOL_TILE_TS 2
#include "smol_k_tile.inc.h"
