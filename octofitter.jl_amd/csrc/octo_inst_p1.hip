// Kernel instantiations and launch code for datasets with 1 planet (see octo_host.h).
#include "octo_launch.h"

namespace octo {
template int dispatch1<1>(octo_ctx*, const octo_dataset*, EvalArgs&, bool, bool, const SmallModel*, hipStream_t);
}
