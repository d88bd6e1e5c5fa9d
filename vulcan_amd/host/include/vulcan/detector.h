// forwarder: Detector is declared in detection.h
#pragma once
#include <vulcan/detection.h>
