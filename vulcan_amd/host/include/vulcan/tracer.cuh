// forwarder: the raycaster is declared in raycast.h
#pragma once
#include <vulcan/raycast.h>
