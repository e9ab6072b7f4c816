// Kernel instantiations and launch code for datasets with 2 planets (see octo_host.h).
#include "octo_launch.h"

namespace octo {
template int dispatch1<2>(octo_ctx*, const octo_dataset*, EvalArgs&, bool, bool, const SmallModel*, hipStream_t);
}
