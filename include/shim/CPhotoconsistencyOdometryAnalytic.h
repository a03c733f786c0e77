// Forwarding header for the reference's unmodified apps (variant A): put this directory
// ahead of the reference's phovo/include on the include path.
#define PHOVO_HIP_USE_REFERENCE_TYPES 1
#include "phovo/CPhotoconsistencyOdometryAnalytic.h"
