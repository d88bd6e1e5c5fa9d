// forwarder: the integrators are declared together in fusion.h
#pragma once
#include <vulcan/fusion.h>
