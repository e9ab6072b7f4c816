// Kernel instantiations and launch code for datasets with 3 planets (see octo_host.h).
#include "octo_launch.h"

namespace octo {
template int dispatch1<3>(octo_ctx*, const octo_dataset*, EvalArgs&, bool, bool, const SmallModel*, hipStream_t);
}
