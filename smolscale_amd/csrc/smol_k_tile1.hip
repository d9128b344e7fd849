/* k_tile instantiations for tap reader 1 (see smol_k_tile.inc.h) */
#define SMOL_TILE_TS 1
#include "smol_k_tile.inc.h"
