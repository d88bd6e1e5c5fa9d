// forwarder: the trackers are declared together in tracking.h
#pragma once
#include <vulcan/tracking.h>
