// irec_team_gang.hip -- the GANG builds of encode_chunk_kernel (irec_team.hip, "Gangs") as a translation unit of their own, so that the
// ten extra instantiations compile beside the product's instead of behind them.  Everything is in irec_team.hip.
#define IREC_TEAM_GANG_TU 1
#include "irec_team.hip"
