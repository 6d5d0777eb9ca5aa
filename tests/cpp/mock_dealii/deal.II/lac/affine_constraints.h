// test scaffolding: see mock_dealii.h
#pragma once
#include "mock_dealii.h"
