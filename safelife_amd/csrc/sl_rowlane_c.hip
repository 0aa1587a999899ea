// Row kernels, third translation unit: the shapes of SL_ROWLANE_SHAPES_C (see the end of sl_rowlane.hip).
#define SL_ROWLANE_PART 2
#include "sl_rowlane.hip"
