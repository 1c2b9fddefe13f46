// irec_team_margin.hip -- the MARGIN builds of encode_team_kernel (irec_team.hip) as a translation unit of their own, so that the
// five extra instantiations compile beside the product's instead of behind them.  Everything is in irec_team.hip.
#define IREC_TEAM_MARGIN_TU 1
#include "irec_team.hip"
