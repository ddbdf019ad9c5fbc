// wire.hip - the reference's packet formats on the host side: wire.h carries the implementation
#include "wire.h"
