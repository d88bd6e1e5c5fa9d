// forwarder: Volume and its flag enums are declared in tsdf_volume.h
#pragma once
#include <vulcan/tsdf_volume.h>
