// Row kernels, second translation unit: the shapes of SL_ROWLANE_SHAPES_B (see the end of sl_rowlane.hip).
#define SL_ROWLANE_PART 1
#include "sl_rowlane.hip"
