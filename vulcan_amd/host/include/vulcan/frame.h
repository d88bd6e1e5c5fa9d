// forwarder: Frame and its image operators are declared in observation.h
#pragma once
#include <vulcan/observation.h>
